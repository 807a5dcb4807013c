timeout 200 python -m pytest tests -q -m gpu -k "check_winograd_up4" 2>&1 | tail -3
for v in "" "--lib tools/lab/ctw_spread.so" "--lib tools/lab/ctw_nowsb.so"; do echo "== $v"; timeout 200 python tools/up4lab.py --frames 64 --reps 5 $v 2>&1 | grep -v amdgpu | tail -4; done
timeout 200 python tools/up4lab.py --frames 16 --ts --lib tools/lab/ctw_ts.so 2>&1 | grep -v amdgpu | grep "\[ts\]"
