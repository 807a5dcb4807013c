"""Lab: which parameter gradients of the 512x512 full-width training graph miss the oracle, with the launch-fusing switches on / off."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as gc
from ipercore_amd.networks import training as tr
for name, sw in (("default", {}), ("no convT wgrad fusion", {"FUSED_CONVT_WGRAD": False}), ("no fused bias", {"FUSED_BIAS_GRAD": False}),
                 ("no convT fwd fusion", {"FUSED_CONVT_FWD": False})):
    prev = {k: getattr(tr, k) for k in sw}
    for k, v in sw.items():
        setattr(tr, k, v)
    try:
        m = gc._generator_training_grads(512, *gc.FULL, real_flows=True)
        print(name, "OK worst", m["worst_rel_grad_err"], m["worst_param"], flush=True)
    except AssertionError as e:
        m = e.args[0] if e.args and isinstance(e.args[0], dict) else {"err": str(e)[:600]}
        print(name, "FAIL", json.dumps({k: m.get(k) for k in ("worst_rel_grad_err", "worst_param", "params_over_tol", "err")}), flush=True)
    for k, v in prev.items():
        setattr(tr, k, v)
