#!/usr/bin/env python3
"""Per-kernel microbenchmarks on the MI355X (c3 shapes by default): time with HIP events over many
launches, report GB/s against algorithmic bytes or TFLOP/s.  Development aid; bench.py is the contract."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdlrm_amd import ops, synth  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8192)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--only", default="")
    ap.add_argument("--alpha", type=float, default=1.05)
    ap.add_argument("--json", default="", help="--only vendor: write the table here")
    a = ap.parse_args()
    B, D = a.B, a.D
    only = set(a.only.split(",")) if a.only else None
    want = lambda n: only is None or n in only

    if want("emb"):
        ln_emb = synth.TERABYTE_COUNTS
        T = len(ln_emb)
        P = 150001
        cs = [min(n, P) for n in ln_emb]
        ctx = ops.CacheCtx(ln_emb, cs, D, 16, B, DEV)
        tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
        weight = torch.randn(ctx.total_rows, D, device=DEV)
        ctx.bind_cache(tags, weight)
        hostbuf = torch.zeros(64, D).pin_memory()
        ctx.bind_host_tables([hostbuf.data_ptr()] * T)
        # resident tags: fill every set/way with an id that maps to it, so the probe hits
        for k in range(T):
            Pk = cs[k]
            ids = torch.arange(Pk * 16, device=DEV) % ln_emb[k]
            sets = ids % Pk
            # way-major fill: way w of set s holds s + w*Pk when that id exists
            w = torch.arange(16, device=DEV).view(1, 16)
            s = torch.arange(Pk, device=DEV).view(Pk, 1)
            cand = s + w * Pk
            cand = torch.where(cand < ln_emb[k], cand, torch.full_like(cand, -1))
            tags[ctx.tag_base[k]:ctx.tag_base[k + 1]] = cand.reshape(-1)
        syn = synth.CriteoSynth(ln_emb, 13, B, alpha=a.alpha, device=DEV)
        idx = syn.window(0, 1)
        for k in range(T):      # restrict to resident ids
            idx[k] %= min(ln_emb[k], cs[k] * 16)
        slots, mp, mc = ops.embbag_probe(ctx, idx)
        torch.cuda.synchronize()
        print("misses:", int(mc.sum()), " distinct slots/lookups: %.3f" % (
            sum(int(torch.unique(slots[k]).numel()) for k in range(T)) / (T * B)))
        feat = torch.empty(B, T + 1, D, device=DEV)
        grad = torch.randn(B, T + 1, D, device=DEV)
        work = ops.embbag_bwd_work(ctx, B, DEV)
        touched = torch.zeros(ctx.total_rows, dtype=torch.uint8, device=DEV)
        look = B * T
        us = timeit(lambda: ops.embbag_probe(ctx, idx))
        print("probe+resolve+fill   %8.1f us   %7.1f GB/s (idx 8B + tags 128B per lookup)" % (us, look * 136 / us / 1e3))
        us = timeit(lambda: ops.embbag_fwd(ctx, slots, None, feat[:, 1:, :], (T + 1) * D, D))
        print("embbag_fwd (gather)  %8.1f us   %7.1f GB/s algorithmic (8D+16 B/lookup) = %.1f%% of 8 TB/s" % (
            us, look * (8 * D + 16) / us / 1e3, look * (8 * D + 16) / us / 1e3 / 80))
        us = timeit(lambda: ops.embbag_bwd_sgd(ctx, slots, None, grad[:, 1:, :], (T + 1) * D, D, 0.01, work, touched))
        print("embbag_bwd_sgd       %8.1f us   %7.1f GB/s algorithmic (12D+8 B/lookup)" % (us, look * (12 * D + 8) / us / 1e3))
        us_p = timeit(lambda: ops.embbag_bwd_prepare(ctx, slots, work))
        us_a = timeit(lambda: ops.embbag_bwd_apply(ctx, B, None, grad[:, 1:, :], (T + 1) * D, D, 0.01, work, touched))
        print("  prepare (slot sort) %7.1f us | apply (sums + row update) %7.1f us = %.1f GB/s = %.1f%% of 8 TB/s" % (
            us_p, us_a, look * (12 * D + 8) / us_a / 1e3, look * (12 * D + 8) / us_a / 1e3 / 80))
        # the apply by kernel form (cdlrm_debug_set 6 bit 64: a lane group per block of 32 sorted positions instead of per position) and by
        # workgroups per CU (key 1); both forms must leave the same rows behind
        from cdlrm_amd import _lib
        w0 = weight.clone()
        res = {}
        for form in (0, 64):
            _lib.raw().cdlrm_debug_set(6, form)
            weight.copy_(w0)
            ops.embbag_bwd_sgd(ctx, slots, None, grad[:, 1:, :], (T + 1) * D, D, 0.01, work, touched)
            res[form] = weight.clone()
            for cap in (0, 4, 24, -1):
                _lib.raw().cdlrm_debug_set(1, cap)
                us_a = timeit(lambda: ops.embbag_bwd_apply(ctx, B, None, grad[:, 1:, :], (T + 1) * D, D, 0.01, work, touched))
                print("  apply form %2d  cap %2d per CU: %7.1f us" % (form, cap, us_a))
            _lib.raw().cdlrm_debug_set(1, 0)
        _lib.raw().cdlrm_debug_set(6, 0)
        print("  block form (64) == position form (0), bitwise:", bool(torch.equal(res[0], res[64])))
        # what cdlrm_embbag_bwd_apply_rest is left with once the interaction backward has done the once-only slots
        for cap in (0, 4, 24, -1):
            _lib.raw().cdlrm_debug_set(1, cap)
            us_r = timeit(lambda: ops.embbag_bwd_apply_rest(ctx, B, None, grad[:, 1:, :], (T + 1) * D, D, 0.01, work, touched))
            print("  apply_rest (runs of >= 2 lookups)  cap %2d per CU: %7.1f us" % (cap, us_r))
        _lib.raw().cdlrm_debug_set(1, 0)

    if want("gather_beside"):
        # how much does the gather (the roofline kernel) lose when it runs beside MFMA-bound GEMMs on another stream?
        ln_emb = synth.TERABYTE_COUNTS
        T = len(ln_emb)
        P = 150001
        cs = [min(n, P) for n in ln_emb]
        ctx = ops.CacheCtx(ln_emb, cs, D, 16, B, DEV)
        tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
        weight = torch.randn(ctx.total_rows, D, device=DEV)
        ctx.bind_cache(tags, weight)
        syn = synth.CriteoSynth(ln_emb, 13, B, alpha=a.alpha, device=DEV)
        idx = syn.window(0, 1)
        slots = torch.stack([(idx[k] % (cs[k] * 16)).to(torch.int32) for k in range(T)]).contiguous()
        feat = torch.empty(B, T + 1, D, device=DEV)
        X = torch.randn(B, 512, device=DEV)
        W = torch.randn(512, 512, device=DEV) / 22.0
        b = torch.randn(512, device=DEV)
        Y = torch.empty(B, 512, device=DEV)
        dW, db = torch.empty(512, 512, device=DEV), torch.empty(512, device=DEV)
        work = ops.linear_bwd_work(B, 512, 512, DEV)
        for prio in (0, -1):
            gs = torch.cuda.Stream(priority=prio)
            for mode in ("alone", "beside forward GEMMs", "beside weight-gradient GEMMs"):
                us = []
                for rep in range(12):
                    e0, e1 = ops.TimingEvent(), ops.TimingEvent()
                    torch.cuda.synchronize()
                    if mode != "alone":
                        for _ in range(6):
                            if mode.startswith("beside forward"):
                                ops.linear_fwd(X, W, b, Y, 1)
                            else:
                                ops.linear_bwd(X, W, None, Y, None, dW, db, 0, work)
                    gs.wait_stream(torch.cuda.current_stream())
                    if mode != "alone":
                        # the GEMM stream keeps going: more GEMMs queued behind the fork
                        for _ in range(6):
                            if mode.startswith("beside forward"):
                                ops.linear_fwd(X, W, b, Y, 1)
                            else:
                                ops.linear_bwd(X, W, None, Y, None, dW, db, 0, work)
                    ops.time_next_gather(ctx, e0, e1)
                    ops.embbag_fwd(ctx, slots, None, feat[:, 1:, :], (T + 1) * D, D, stream=gs)
                    torch.cuda.synchronize()
                    if rep >= 2:
                        us.append(e0.elapsed_us(e1))
                print("gather stream priority %2d, %-30s: %6.1f us (min %.1f max %.1f)" % (prio, mode, np.mean(us), min(us), max(us)))

    if want("fused"):
        # gather + interaction as ONE launch (cdlrm_gather_interact_fwd / _bwd) against the two operators: bit-identical results,
        # time by workgroups per CU (cdlrm_debug_set 4 / 5)
        from cdlrm_amd import _lib
        ln_emb = synth.TERABYTE_COUNTS
        T = len(ln_emb)
        P = 150001
        cs = [min(n, P) for n in ln_emb]
        ctx = ops.CacheCtx(ln_emb, cs, D, 16, B, DEV)
        tags = torch.full((ctx.total_tags,), -1, dtype=torch.int64, device=DEV)
        weight = torch.randn(ctx.total_rows, D, device=DEV)
        ctx.bind_cache(tags, weight)
        syn = synth.CriteoSynth(ln_emb, 13, B, alpha=a.alpha, device=DEV)
        idx = syn.window(0, 1)
        slots = torch.stack([(idx[k] % (cs[k] * 16)).to(torch.int32) for k in range(T)]).contiguous()
        F = T + 1
        npairs = F * (F - 1) // 2
        ld = (D + npairs + 3) // 4 * 4
        feat = torch.zeros(B, F, D, device=DEV)
        feat[:, 0, :] = torch.randn(B, D, device=DEV).clamp_(min=0)
        xonly = torch.zeros(B, F, D, device=DEV)          # the fused kernels' operand: feature 0 alone, rows 1.. poisoned
        xonly[:, 0, :] = feat[:, 0, :]
        xonly[:, 1:, :] = float("nan")
        R, R2 = torch.zeros(B, ld, device=DEV), torch.zeros(B, ld, device=DEV)
        dR = torch.randn(B, ld, device=DEV)
        dfeat, dfeat2 = torch.zeros(B, F, D, device=DEV), torch.zeros(B, F, D, device=DEV)
        ops.embbag_fwd(ctx, slots, None, feat[:, 1:, :], F * D, D)
        ops.interact_fwd(feat, False, R)
        ops.gather_interact_fwd(ctx, slots, xonly[:, 0, :], False, R2)
        ops.interact_bwd(feat, dR, False, dfeat, x_act=1)
        ops.gather_interact_bwd(ctx, slots, xonly[:, 0, :], dR, False, dfeat2, x_act=1)
        torch.cuda.synchronize()
        print("fused forward  == gather + interact_fwd bit for bit:", bool(torch.equal(R, R2)))
        print("fused backward == interact_bwd on the gathered block:", bool(torch.equal(dfeat, dfeat2)))
        look = B * T
        us_g = timeit(lambda: ops.embbag_fwd(ctx, slots, None, feat[:, 1:, :], F * D, D))
        us_i = timeit(lambda: ops.interact_fwd(feat, False, R))
        us_b = timeit(lambda: ops.interact_bwd(feat, dR, False, dfeat, x_act=1))
        print("gather %.1f us + interact_fwd %.1f us = %.1f us | interact_bwd %.1f us" % (us_g, us_i, us_g + us_i, us_b))
        fbytes = look * (4 * D + 4) + B * 4 * D + B * 4 * ld      # rows + slot ids + dense feature + output rows
        for k in (1, 2, 3, 4):
            _lib.raw().cdlrm_debug_set(4, k)
            us = timeit(lambda: ops.gather_interact_fwd(ctx, slots, xonly[:, 0, :], False, R2))
            print("fused forward,  %d workgroups per CU: %6.1f us  %.2f TB/s of its own bytes (%.1f MB)" % (k, us, fbytes / us / 1e6, fbytes / 1e6))
        _lib.raw().cdlrm_debug_set(4, 0)
        for k in (1, 2):
            _lib.raw().cdlrm_debug_set(5, k)
            us = timeit(lambda: ops.gather_interact_bwd(ctx, slots, xonly[:, 0, :], dR, False, dfeat2, x_act=1))
            print("fused backward, %d workgroups per CU: %6.1f us" % (k, us))
        _lib.raw().cdlrm_debug_set(5, 0)

    if want("interact_beside"):
        # the interaction forward (HBM-bound, on the critical path) beside a weight-gradient GEMM on another stream: how much
        # does each lose?  (Would deferring the last top-MLP weight gradient under the next step's interaction pay?)
        F = 27
        feat = torch.randn(B, F, D, device=DEV)
        npairs = F * (F - 1) // 2
        R = torch.empty(B, (D + npairs + 3) // 4 * 4, device=DEV)
        X = torch.randn(B, 512, device=DEV)
        dZ = torch.randn(B, 256, device=DEV)
        W = torch.randn(256, 512, device=DEV) / 22.0
        dW, db = torch.empty(256, 512, device=DEV), torch.empty(256, device=DEV)
        work = ops.linear_bwd_work(B, 256, 512, DEV)
        gs = torch.cuda.Stream()
        def ev_time(fn, stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fn(); e1.record(stream)
            return e0, e1
        for mode in ("alone", "together"):
            ti, tg = [], []
            for rep in range(14):
                torch.cuda.synchronize()
                main = torch.cuda.current_stream()
                gs.wait_stream(main)
                if mode == "together":
                    g0, g1 = ev_time(lambda: ops.linear_bwd(X, W, None, dZ, None, dW, db, 0, work, stream=gs), gs)
                i0, i1 = ev_time(lambda: ops.interact_fwd(feat, False, R), main)
                if mode == "alone":
                    torch.cuda.synchronize()
                    g0, g1 = ev_time(lambda: ops.linear_bwd(X, W, None, dZ, None, dW, db, 0, work, stream=gs), gs)
                torch.cuda.synchronize()
                if rep >= 2:
                    ti.append(i0.elapsed_time(i1) * 1e3); tg.append(g0.elapsed_time(g1) * 1e3)
            print("%-9s interact_fwd %6.1f us   weight gradient 256x512 (+ reduce) %6.1f us" % (mode, np.mean(ti), np.mean(tg)))

    if want("interact"):
        F = 27
        feat = torch.randn(B, F, D, device=DEV)
        npairs = F * (F - 1) // 2
        width = (D + npairs + 3) // 4 * 4          # the engine pads the interaction output to a 16-byte row pitch
        R = torch.empty(B, width, device=DEV)
        dR = torch.randn(B, width, device=DEV)
        dfeat = torch.empty_like(feat)
        us = timeit(lambda: ops.interact_fwd(feat, False, R))
        byt = B * (F * D * 4 + (D + npairs) * 4)
        print("interact_fwd         %8.1f us   %7.1f GB/s   %6.1f TFLOP/s" % (us, byt / us / 1e3, 2 * B * 32 * 32 * D / us / 1e6))
        us = timeit(lambda: ops.interact_bwd(feat, dR, False, dfeat))
        byt = B * (2 * F * D * 4 + (D + npairs) * 4)
        print("interact_bwd         %8.1f us   %7.1f GB/s" % (us, byt / us / 1e3))

    if want("thin"):
        # the layers with a thin side: 13-wide input (forward on the vector ALU, weights in registers), and the weight gradients
        # of the 13-wide and the 1-wide layer (LDS-free MFMA kernel + the grouped slab reduction)
        X = torch.randn(B, 13, device=DEV)
        W = torch.randn(512, 13, device=DEV) / 3.6
        b = torch.randn(512, device=DEV)
        Y = torch.empty(B, 512, device=DEV)
        us = timeit(lambda: ops.linear_fwd(X, W, b, Y, 1))
        print("linear 13->512 fwd     %8.1f us   %7.1f GB/s written" % (us, B * 512 * 4 / us / 1e3))
        for (N, K) in ((512, 13), (1, 256), (256, 512)):
            Xs = [torch.randn(B, K, device=DEV)]
            dZs = [torch.randn(B, N, device=DEV)]
            dWs, dbs = [torch.empty(N, K, device=DEV)], [torch.empty(N, device=DEV)]
            plan = ops.WgradPlan(Xs, dZs, dWs, dbs, ops.mlp_wgrad_work(B, [N], [K], DEV))
            us = timeit(lambda: ops.mlp_wgrad(plan))
            print("wgrad %4d x %4d (+ slab reduction) %8.1f us   %7.1f GB/s of operands" % (N, K, us, B * (N + K) * 4 / us / 1e3))

    if a.only and "vendor" in only:
        # YARDSTICK, tools only, never product: the vendor library's fp32 GEMM (torch.mm / addmm -> hipBLASLt / rocBLAS; the
        # reference hands its MLPs to exactly that, model_no_ddp.py:244-270) on the step's GEMM shapes, same box, same process,
        # beside this repo's kernels.  allow_tf32 stays off: both sides compute true fp32 products.
        import json
        torch.backends.cuda.matmul.allow_tf32 = False
        rows = []
        shapes = [(512, 480), (512, 512), (256, 512), (128, 256)]       # (N = out, K = in) of the LDS-tiled layers at c3
        for M in (8192, 65536, 1024, 2048):
            for N, K in shapes:
                X = torch.randn(M, K, device=DEV)
                W = torch.randn(N, K, device=DEV) / np.sqrt(K)
                b = torch.randn(N, device=DEV)
                Y = torch.empty(M, N, device=DEV)
                dY = torch.randn(M, N, device=DEV)
                dX = torch.empty(M, K, device=DEV)
                dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
                work = ops.linear_bwd_work(M, N, K, DEV)
                plan = ops.WgradPlan([X], [dY], [dW], [db], ops.mlp_wgrad_work(M, [N], [K], DEV))
                Wt = W.t()
                dYt = dY.t()
                fl = 2.0 * M * N * K
                reps = 10 if M > 8192 else 30
                r = dict(M=M, N=N, K=K, gflop=fl / 1e9)
                r["fwd_ours_us"] = timeit(lambda: ops.linear_fwd(X, W, b, Y, 1), reps)                     # bias + ReLU fused
                # ... and as the training step launches its top MLP (CDLRM_GEMM_ALONE: the wide kernel where its tiles fill the chip)
                r["fwd_ours_alone_us"] = timeit(lambda: ops.linear_fwd(X, W, b, Y, 1, alone=True), reps)
                r["dgrad_ours_alone_us"] = timeit(lambda: ops.linear_bwd(X, W, Y, dY, dX, None, None, 0, work, x_act=1, alone=True), reps)
                r["fwd_vendor_us"] = timeit(lambda: torch.addmm(b, X, Wt, out=Y), reps)                    # bias, no activation
                r["fwd_vendor_relu_us"] = timeit(lambda: torch.relu_(torch.addmm(b, X, Wt, out=Y)), reps)
                r["dgrad_ours_us"] = timeit(lambda: ops.linear_bwd(X, W, Y, dY, dX, None, None, 0, work, x_act=1), reps)  # act' fused
                r["dgrad_vendor_us"] = timeit(lambda: torch.mm(dY, W, out=dX), reps)
                r["wgrad_ours_us"] = timeit(lambda: ops.mlp_wgrad(plan), reps)                             # dW + db (+ slab reduce)
                r["wgrad_vendor_us"] = timeit(lambda: torch.mm(dYt, X, out=dW), reps)
                r["wgrad_vendor_db_us"] = timeit(lambda: (torch.mm(dYt, X, out=dW), torch.sum(dY, 0, out=db)), reps)
                for k in ("fwd_ours", "fwd_ours_alone", "fwd_vendor", "dgrad_ours", "dgrad_ours_alone", "dgrad_vendor", "wgrad_ours", "wgrad_vendor"):
                    r[k + "_tflops"] = fl / r[k + "_us"] / 1e6
                rows.append(r)
                print("M=%6d N=%4d K=%4d | fwd ours %7.1f us (%5.1f TF; alone-hint %7.1f) vendor %7.1f (+relu %7.1f) | dgrad ours %7.1f "
                      "(alone-hint %7.1f) vendor %7.1f | "
                      "wgrad ours %7.1f vendor %7.1f (+db %7.1f)" % (M, N, K, r["fwd_ours_us"], r["fwd_ours_tflops"], r["fwd_ours_alone_us"],
                                                                    r["fwd_vendor_us"],
                                                                    r["fwd_vendor_relu_us"], r["dgrad_ours_us"], r["dgrad_ours_alone_us"],
                                                                    r["dgrad_vendor_us"],
                                                                    r["wgrad_ours_us"], r["wgrad_vendor_us"], r["wgrad_vendor_db_us"]),
                      flush=True)
                del X, W, Y, dY, dX, work, plan
        if a.json:
            json.dump({"what": "fp32 GEMMs of the c3 step: this repo's kernels (epilogues fused: bias + ReLU, act', dW + db) against "
                               "torch.mm / addmm = the vendor library, same box, same process, back-to-back launches (HIP events "
                               "around 10-30 launches)", "torch": torch.__version__,
                       "device": torch.cuda.get_device_name(0), "rows": rows}, open(a.json, "w"), indent=1)

    if want("gemm"):
        layers = [(13, 512, 1), (512, 256, 1), (256, 128, 1), (D + 351, 512, 1), (512, 512, 1), (512, 256, 1), (256, 1, 2)]
        tot_f = tot_b = 0.0
        for K, N, act in layers:
            X = torch.randn(B, K, device=DEV)
            W = torch.randn(N, K, device=DEV) / np.sqrt(K)
            b = torch.randn(N, device=DEV)
            Y = torch.empty(B, N, device=DEV)
            dY = torch.randn(B, N, device=DEV)
            dX, dW, db = torch.empty(B, K, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
            work = ops.linear_bwd_work(B, N, K, DEV)
            fl = 2.0 * B * N * K
            uf = timeit(lambda: ops.linear_fwd(X, W, b, Y, act))
            ub = timeit(lambda: ops.linear_bwd(X, W, Y, dY, dX, dW, db, act, work))
            tot_f += uf
            tot_b += ub
            print("linear %4d->%4d  fwd %7.1f us %6.1f TF   bwd %7.1f us %6.1f TF" % (K, N, uf, fl / uf / 1e6, ub, 2 * fl / ub / 1e6))
        print("MLP total fwd %.1f us  bwd %.1f us" % (tot_f, tot_b))


if __name__ == "__main__":
    main()
