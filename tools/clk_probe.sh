cd /root/repo; export TMPDIR=/tmp; R=$PWD
( cd /tmp && LWG_SPLIT_PP=1 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcs_clk -o pmc -- python $R/tools/convlab.py --split --bf16 --iters 20 --shapes res64,skip0 $R/ipercore_amd/liblwg_hip.so > $R/gpurun_out/pmcs_clk.log 2>&1 )
python tools/pmc_summary.py gpurun_out/pmcs_clk | grep -E "kernel|lwg_" | cut -c1-200
