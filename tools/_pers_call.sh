cd /root/repo
timeout 600 python -m pytest tests -q -m gpu -k "check_panel_cache_refresh or check_graph_vs_eager_steps" 2>&1 | tail -5
for a in "" "--use-vgg --use-face"; do timeout 600 python bench_personalize.py --steps 10 --warmup 4 $a 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$a', d['ms_per_step'], d['roofline']['frac'], d['loss_G'], d['loss_D'], d.get('self_check'))"; done
