#!/usr/bin/env python
"""Rehearsal of the block-row distributed GP fit (scasml_gp_amd/dist_gp.py) with `--ranks` processes sharing ONE GPU over gloo
(the RCCL run needs the multi-GPU node): Gram rows, distributed Cholesky, preconditioned Newton-CG, against the single-GPU
GPsolver on the same data.  Prints one JSON line per rank 0.

    python tools/dist_gp_demo.py --ranks 2 --d 250 --n-dom 8333 --n-bdy 1667      # M = 34 999: BASELINE configs[4], staged
    python tools/dist_gp_demo.py --compat none ...                                 # the documented operators instead of the as-coded surrogate
    python tools/dist_gp_demo.py --ranks 1 --backend nccl ...                      # one rank, every collective issued through RCCL
    python tools/dist_gp_demo.py --n-dom 16667 --n-bdy 3333 --factor-only          # M = 70 001 (39 GB): past 2^31 matrix elements
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, args, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if args.backend == "nccl":        # RCCL: one rank per GPU; on a one-GPU box that is ONE rank, with every collective forced through the backend
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from scasml_gp_amd.dist_gp import Comm, DistCholesky, DistributedGP
    force = args.backend == "nccl" and world == 1
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(args.d + 1)
    np.random.seed(1234)
    dom, bdy = eq.generate_data(args.n_dom, args.n_bdy)
    xt = np.concatenate(eq.generate_test_data(500, 100))
    compat = None if args.compat == "none" else "reference"
    out = {"ranks": world, "backend": dist.get_backend(), "collectives_forced_at_one_rank": force, "d": args.d, "collocation": "%d+%d" % (args.n_dom, args.n_bdy), "surrogate": "as coded (compat='reference')" if compat else "documented operators"}
    say = lambda msg: print("[rank %d, %.0f s] %s" % (rank, time.perf_counter() - t_start, msg), file=sys.stderr, flush=True) if rank == 0 else None
    t_start = time.perf_counter()
    probe = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
    cm = Comm(force=force)
    # stage timings of the distributed factorisation alone
    if not args.fit_only:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ch = DistCholesky(args.d, 1.0 / (0.25 ** 2 * args.d), dom, bdy, 1e-2, cm, compat_idx=probe.laplacian_idx).build()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        say("Gram rows built")
        ch.factor()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        say("factored")
        b = torch.from_numpy(np.random.default_rng(0).standard_normal(ch.M)).cuda()
        ch.solve(b)                                            # the first solve also assembles and inverts the diagonal super-blocks (once per factor)
        torch.cuda.synchronize(); t_first = time.perf_counter() - t2
        torch.cuda.synchronize(); ta = time.perf_counter()
        ch.solve(b)
        torch.cuda.synchronize(); tb = time.perf_counter()
        ch.matvec(b)                                           # (allocates the ordered sweep's scratch)
        torch.cuda.synchronize(); tc = time.perf_counter()
        ch.matvec(b)
        torch.cuda.synchronize(); td = time.perf_counter()
        t2, t3, t4 = t2, t2 + (tb - ta), t2 + (tb - ta) + (td - tc)
        out["first_solve_s_incl_group_inverses"] = round(t_first, 3)
        out.update(M=ch.M, block_rows=ch.nblk, panel_gb_per_rank=round(ch.memory_bytes() / 1e9, 2), gram_s=round(t1 - t0, 3),
                   factor_s=round(t2 - t1, 3), factor_tflops_all_ranks=round(ch.M ** 3 / 3 / (t2 - t1) / 1e12, 2),
                   solve_s=round(t3 - t2, 3), matvec_s=round(t4 - t3, 3), collective_gb_per_rank=round(cm.bytes_moved / 1e9, 2),
                   collective_calls=dict(cm.calls),
                   gram_pair_rows_per_s=round(sum(min(256, ch.M - i * 256) for i in ch.mine) * (args.n_dom + args.n_bdy) / (t1 - t0), 1))
        del ch
        torch.cuda.empty_cache()
        if args.factor_only:
            if rank == 0:
                q.put(out)
            dist.barrier()
            dist.destroy_process_group()
            return
    gp = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
    fit = DistributedGP(gp, Comm(force=force))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    say("distributed fit starts")
    fit.fit(dom, bdy, GN_steps=20, progress=say, cg_tol="adaptive" if args.adaptive_cg else 1e-10)
    out["cg_tol"] = "adaptive (inexact Newton)" if args.adaptive_cg else 1e-10
    torch.cuda.synchronize(); t1 = time.perf_counter()
    say("distributed fit done")
    out.update(fit_s=round(t1 - t0, 2), newton_steps=len(gp.loss_history) - 1, cg_products=fit.cg_iterations,
               loss=[float("%.6g" % v) for v in gp.loss_history], grad_norm_last=gp.grad_norms[-1])
    pred = gp.predict(xt)
    exact = eq.exact_solution(xt)
    out["gp_rel_l2"] = round(float(np.linalg.norm(pred - exact) / np.linalg.norm(exact)), 4)
    if rank == 0 and not args.no_single:
        one = GP_Grad_Dependent_Nonlinear(eq, compat=compat)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        one.GPsolver(dom, bdy, GN_steps=20)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.update(single_gpu_fit_s=round(t1 - t0, 2), single_gpu_newton_steps=len(one.loss_history) - 1,
                   right_vector_rel_diff=float(np.abs(gp.right_vector - one.right_vector).max() / np.abs(one.right_vector).max()),
                   predict_max_diff=float(np.abs(pred - one.predict(xt)).max()))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--d", type=int, default=250)
    ap.add_argument("--n-dom", type=int, default=8333)
    ap.add_argument("--n-bdy", type=int, default=1667)
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--adaptive-cg", action="store_true", help="inexact Newton: the inner CG tolerance follows the gradient norm (DistributedGP.fit(cg_tol='adaptive'))")
    ap.add_argument("--factor-only", action="store_true", help="Gram rows, factorisation, one solve and one matvec; no Newton fit")
    ap.add_argument("--fit-only", action="store_true", help="skip the stage timings before the fit (sizes whose panel fills more than a third of the GPU: "
                                                             "the caching allocator may still hold the first panel when the fit allocates its own)")
    ap.add_argument("--backend", choices=["gloo", "nccl"], default="gloo",
                    help="nccl = RCCL: needs one GPU per rank, so --ranks 1 on a one-GPU box (every collective is then forced through RCCL: Comm(force=True))")
    ap.add_argument("--compat", choices=["reference", "none"], default="reference",
                    help="reference (default): the as-coded surrogate -- Gram rows of scasml_gp_gram_compat_rows, the float16-rounded matrix solved by a second "
                         "distributed factorisation; none: the documented operators")
    args = ap.parse_args()
    if args.backend == "nccl" and args.ranks != 1:
        raise SystemExit("--backend nccl on this one-GPU tool needs --ranks 1 (RCCL refuses two ranks on one device)")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.ranks, port, args, q)) for r in range(args.ranks)]
    for p in procs:
        p.start()
    import queue as _queue
    result = None
    while result is None:                                 # a worker that dies (out of memory, a failed collective) must not leave this process waiting
        try:
            result = q.get(timeout=5)
        except _queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise SystemExit("a rank exited with an error before the result was reported")
    print(json.dumps(result), flush=True)
    for p in procs:
        p.join()


if __name__ == "__main__":
    main()
