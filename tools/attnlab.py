"""Lab: the attention block (lwg_lwb_attention_x_*) at the launch shapes bench.py runs, with rendered flows.
usage: attnlab.py [LIB.so ...]   (each library timed in its own process: the product library reads no environment)"""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(lib):
    import torch
    from ipercore_amd import _lib
    if lib != "product":
        _lib.LIB_PATH = os.path.abspath(lib)
    from ipercore_amd import ops, synthetic as syn
    dev = "cuda:0"
    res = {}
    for S, FB, dt, workload in ((512, 32, torch.float32, "imitate"), (1024, 20, torch.bfloat16, "novel")):
        case = syn.build_case(image_size=S, n_frames=FB, ns=2)
        im = syn.make_imitator(case, frame_batch=FB, device=dev)
        if workload == "novel":
            import bench
            case.tgt_smpls = bench.novel_view_smpls(case, im.body_rec.np_hands_mean, 180)[::9][:FB]
        tgt = im.prepare_sequence(case.tgt_smpls, "smooth")
        _, Tst, _ = im.make_inputs_for_tsf(im.src_info, tgt[:FB], "smooth", t=0)
        del im
        for h, C in ((S // 8, 256), (S // 4, 128), (S // 2, 64)):
            T = ops.flow_resize(Tst, h, h)
            fg = ((T[..., 0] > -1.5) & (T[..., 0] < 1.5)).float().mean().item()
            x = torch.randn(FB, h, h, C, device=dev).to(dt)
            Kq, Vs = (torch.randn(2, h, h, C, device=dev).to(dt) for _ in range(2))
            kap, bv = torch.randn(2, h, h, device=dev), torch.randn(C, device=dev)
            out = torch.empty_like(x)
            nrec = ops.attn_records(h, h, C, x.dtype)
            ws = torch.empty(ops.instnorm_finalize_ws(FB, C, nrec), device=dev)
            mean, rstd = torch.empty(FB, C, device=dev), torch.empty(FB, C, device=dev)
            for _ in range(3):
                ops.lwb_attention_x(x, Kq, kap, Vs, bv, T, out, stats=ws)
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            n = 20
            e0.record()
            for _ in range(n):
                ops.lwb_attention_x(x, Kq, kap, Vs, bv, T, out, stats=ws)
            e1.record()
            for _ in range(n):
                ops.instnorm_finalize(ws, FB, C, nrec, mean, rstd)
            e2.record()
            torch.cuda.synchronize()
            us, us_m = e0.elapsed_time(e1) / n * 1e3, e1.elapsed_time(e2) / n * 1e3
            nbytes = 2 * x.numel() * x.element_size()
            lat = {}
            for nb in (1, 2):                       # a few frames: the sixteen-wave form
                xs, Ts, outs = x[:nb].contiguous(), T[:nb].contiguous(), out[:nb].contiguous()
                for _ in range(3):
                    ops.lwb_attention_x(xs, Kq, kap, Vs, bv, Ts, outs, stats=ws)
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(n):
                    ops.lwb_attention_x(xs, Kq, kap, Vs, bv, Ts, outs, stats=ws)
                f1.record()
                torch.cuda.synchronize()
                lat[f"us_B{nb}"] = round(f0.elapsed_time(f1) / n * 1e3, 1)
            res[f"{'f32' if dt == torch.float32 else 'bf16'} {FB}x{h}x{h}x{C}"] = {"us": round(us, 1), "TB/s x+out": round(nbytes / us / 1e6, 2), "merge_us": round(us_m, 1),
                                                                                      "fg_frac": round(fg, 3), "nrec": nrec, **lat}
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2])
    else:
        for lib in (sys.argv[1:] or ["product"]):
            r = subprocess.run([sys.executable, __file__, "--worker", lib], capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            print("==", lib)
            if not line:
                print(r.stdout[-1500:], r.stderr[-3000:])
                continue
            for k, v in json.loads(line[0][7:]).items():
                print(f"  {k:28s} {v}")
