#!/usr/bin/env python3
"""Run every GPU parity check, never stop at the first failure, write gpurun_out/diag.json (+ traceback text)."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tests import gpu_checks
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    only = set(sys.argv[1:])
    report = {"device": torch.cuda.get_device_name(0), "checks": {}}
    for fn in gpu_checks.ALL:
        if only and fn.__name__ not in only:
            continue
        t0 = time.time()
        try:
            report["checks"][fn.__name__] = {"ok": True, "metrics": fn()}
        except Exception as e:   # noqa: BLE001
            report["checks"][fn.__name__] = {"ok": False, "error": f"{type(e).__name__}: {e}"[:2000],
                                             "trace": traceback.format_exc()[-3000:]}
        report["checks"][fn.__name__]["seconds"] = round(time.time() - t0, 2)
        print(fn.__name__, "OK" if report["checks"][fn.__name__]["ok"] else "FAIL " + report["checks"][fn.__name__]["error"][:300],
              flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as fp:
            json.dump(report, fp, indent=1, default=str)
    bad = [k for k, v in report["checks"].items() if not v["ok"]]
    print("FAILED:", bad if bad else "none")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
