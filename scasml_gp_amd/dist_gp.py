"""Block-row distributed Gram / Cholesky / solves and the matrix-free Newton fit on top of them, for collocation sets
whose feature Gram does not fit one GPU (BASELINE.json configs[4]: d = 250, 1e5 collocation points -> M = 350 000 features,
K(phi, phi) = 980 GB float64).  Replaces models/GP.py:182-268 (kernel_phi_phi + factor), :487-604 (GPsolver) and the solve
of :599 at that size.  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI; "gloo" in rehearsals and tests).

Layout.  K is cut into block rows of B = 256 feature rows; block row i belongs to rank i % world (1-D block-cyclic, so every
rank holds the same share of the lower triangle) and a rank stores its block rows stacked in ONE row-major panel
``R`` of (owned * 256) x Mp float64 (Mp = M rounded up to 256, identity padding).  Only columns <= the block's own are ever
touched.  Memory at M = 350 000 on 8 GPUs: 171 block rows x 256 x 350 208 x 8 B = 122.6 GB per GPU (61 GB of it is the
lower triangle) + 0.7 GB of replicated diagonal blocks + 0.7 GB for the broadcast panel of a step: fits the 288 GB of a MI355X
more than twice over; the same fit on 4 GPUs (245 GB) still fits.

Right-looking factorisation, per block column k (1367 steps at M = 350 000):
  1. owner(k) factors its 256 x 256 diagonal block (scasml_cholesky) and BROADCASTS it (512 KB);
  2. every rank solves its blocks of column k against it (scasml_trsm_right_lt on one strided view of R: its block rows > k
     are a suffix of the stack);
  3. ONE all-gather assembles the column panel (all blocks (i, k), i > k, in global order: (nb - k - 1) * 512 KB);
  4. every rank updates its trailing block rows with ONE launch: R[suffix, k+1:] -= P_mine P_all^T on the FP64 matrix cores,
     tiles above the block diagonal skipped (scasml_gemm_nt_sub with its triangular map).
Block columns are eliminated in PAIRS (factor(pair=True), the default): steps 1-3 for column k, step 4 for block column k + 1 alone, steps
1-3 for column k + 1, then ONE trailing update with K = 512 by both panels -- the same collectives, half the trailing launches, and the
128 x 128 update tile's fixed cost (the read-modify-write of C) paid once per 512 columns of K.
Collective volume: the panel of step k reaches every rank once, M^2 / 2 * 8 B = 490 GB per rank over the whole run -- over
xGMI rings (7 links x ~50 GB/s effective each way) a few seconds against ~60 s of FP64 MFMA work (M^3 / 3 = 1.4e16 flop / 8 GPUs).

Substitutions (K_p^-1 z is two of them; the explicit inverse the single-GPU path forms is never built) walk the block rows in GROUPS of
four (1024 rows): the diagonal SUPER-block L_GG of a group is assembled on every rank once after factor() (one all-reduce of 8 MB per
group) and inverted, so a group costs ONE collective of 1024 doubles and three launches instead of four collectives and ~16 launches:
forward -- the owners' up-to-date pieces of b_G meet in one all-reduce, y_G = L_GG^-1 b_G is one 1024 x 1024 product on every rank,
one gemv takes L_{>G,G} y_G off the rank's later rows; backward -- one all-reduce sums the ranks' accumulators for the group's
columns, x_G = L_GG^-T t_G, and the group's owners fold x_G into their accumulators over the columns before the group.  A solve is
2 ceil(nblk / 4) collectives (M = 350 000: 684 instead of 2 736) and reads the triangle twice.

Newton (models/GP.py:501-588) without K_p^-1 and without the (3 N_dom)^2 Hessian (500 GB at this size): the damped Newton
system (H + 1e-4 I) delta = -g is solved INEXACTLY by conjugate gradients with H v = 2 J^T K_p^-1 J v + second-derivative
term (scasml_gp_newton_jv / _jtv), one distributed solve per product; where CG meets negative curvature the step falls back
to the Gauss-Newton operator, as the single-GPU path does.  H inherits the conditioning of K_p (1e5-1e6 at the
reference's sizes); CG is preconditioned with P = 1/2 S K_p S^T, S the selector of the (z1, z3, z5) rows of b -- exact
for the rows on which b is the identity in sol, i.e. everything but the F rows: cond 1.5e5 -> 73 and 1900 -> 35 iterations at
300 + 60 points (measured on the oracle).  P v is one product with K_p = L L^T: two local gemv sweeps over the rank's block
rows and TWO collectives of an M-vector (DistCholesky.matvec), against two collectives PER BLOCK for a solve.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

BLK = _lib.DIST_BLOCK
GROUP = 4          # block rows per substitution step (1024 rows: the transposed sweeps stay one-writer-per-column, i.e. bitwise reproducible)


class Comm:
    """The three collectives the path uses, on CUDA tensors; gloo groups (rehearsal on one GPU, tests) go through the host.

    ``force`` (or SCASML_DIST_FORCE_COLLECTIVES=1): issue every collective at world == 1 too.  A one-rank group makes each of them a copy
    onto itself, but the call goes through the backend -- on a one-GPU box this is the only way the RCCL branches (device tensors, issued
    from the two streams of the look-ahead factorisation) execute at all (tests/test_gpu_dist_gp.py, tools/dist_gp_demo.py --backend nccl)."""

    def __init__(self, group=None, force=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.backend = dist.get_backend(group) if self.on else None
        self.host = self.backend == "gloo"
        if force is None:
            force = os.environ.get("SCASML_DIST_FORCE_COLLECTIVES") == "1"
        self.active = self.world > 1 or (bool(force) and self.on)
        self.bytes_moved = 0
        self.calls = {"broadcast": 0, "all_reduce": 0, "all_gather": 0}

    def _src(self, src):
        return src if self.group is None else self.dist.get_global_rank(self.group, src)

    def _via(self, t, fn):
        if not self.host:
            fn(t)
            return t
        h = t.cpu()
        fn(h)
        t.copy_(h)
        return t

    def broadcast(self, t, src):
        if self.active:
            self.calls["broadcast"] += 1
            self.bytes_moved += t.numel() * t.element_size()
            self._via(t, lambda x: self.dist.broadcast(x, src=self._src(src), group=self.group))
        return t

    def all_reduce(self, t):
        if self.active:
            self.calls["all_reduce"] += 1
            self.bytes_moved += 2 * t.numel() * t.element_size()
            self._via(t, lambda x: self.dist.all_reduce(x, group=self.group))
        return t

    def all_gather(self, t):
        """-> (world, *t.shape) tensor; every rank passes the same shape."""
        import torch
        if not self.active:
            return t.unsqueeze(0)
        self.calls["all_gather"] += 1
        self.bytes_moved += self.world * t.numel() * t.element_size()
        if self.host:
            # gloo's all_gather moves ~0.3 GB/s between two local ranks, its broadcast ~3 GB/s (measured on this image): the rehearsal gathers
            # with one broadcast per rank into a single host buffer
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype)
            out[self.rank].copy_(t)
            for q in range(self.world):
                self.dist.broadcast(out[q], src=self._src(q), group=self.group)
            return out.to(t.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out


def owned_blocks(nblk, rank, world):
    """Global block-row indices of ``rank`` (1-D block-cyclic)."""
    return list(range(rank, nblk, world))


class DistCholesky:
    """K(phi, phi) + nugget I, block-row distributed: build(), factor(), solve(b)."""

    def __init__(self, d, a, x_dom, x_bdy, nugget, comm=None, compat_idx=None, round_diag=False, f16_graph=False, f16_extra=0):
        """compat_idx: the five Hutchinson indices -> the Gram AS CODED by the reference (shifted blocks, float16 entries:
        scasml_gp_gram_compat_rows); round_diag: the diagonal of K + nugget I rounded to float16 as well, i.e. the matrix
        kernel_phi_phi_perturb.astype(float16) of models/GP.py:268 that the right_vector solve of :599 uses."""
        torch = _lib.require_gpu()
        self.compat_idx = None if compat_idx is None else np.ascontiguousarray(np.asarray(compat_idx, dtype=np.int32))
        self.round_diag = bool(round_diag)
        # scasml_gp_gram_compat_rows: bit 2 = the float16 op sequence on float16 rows (GP(f16_graph=True)); f16_extra = the GP's exploratory bit 3 --
        # the same mask GP.kernel_phi_phi builds, so the distributed and the single-GPU Gram cannot differ
        self.gram_bits = 1 | ((4 | int(f16_extra)) if f16_graph else 0)
        self.lib = _lib.load()
        self.comm = comm or Comm()
        self.d, self.a, self.nugget = int(d), float(a), float(nugget)
        self.xd = torch.from_numpy(np.ascontiguousarray(np.asarray(x_dom), dtype=np.float32)).cuda()
        self.xb = torch.from_numpy(np.ascontiguousarray(np.asarray(x_bdy), dtype=np.float32)).cuda()
        self.n_dom, self.n_bdy = self.xd.shape[0], self.xb.shape[0]
        self.M = 4 * self.n_dom + self.n_bdy
        self.nblk = (self.M + BLK - 1) // BLK
        self.Mp = self.nblk * BLK
        self.mine = owned_blocks(self.nblk, self.comm.rank, self.comm.world)
        self.R = None
        self.diag = [None] * self.nblk          # replicated 256 x 256 diagonal factors
        self.ninv = None                        # -L_GG^-1 of every group of GROUP diagonal blocks, made by the first solve() (_neg_inverse_groups)
        self._scratch = None                    # partial sums of the ordered transposed sweep (matvec)
        self._mine_dev = None
        self.info = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.bad = torch.zeros(1, dtype=torch.float64, device="cuda")   # blocks this rank found not positive definite (scasml_cholesky resets info per call)

    # -------------------------------------------------------------------------------------------- helpers
    def _slot0(self, k):
        """First local slot whose global block row is > k (the owned block rows > k are a suffix of the stack)."""
        r, w = self.comm.rank, self.comm.world
        return 0 if k < r else (k - r) // w + 1

    def _ptr(self, t, row, col, ld):
        return C.c_void_p(t.data_ptr() + 8 * (row * ld + col))

    def memory_bytes(self):
        return len(self.mine) * BLK * self.Mp * 8

    @staticmethod
    def budget(d, n_dom, n_bdy, world, rank=0):
        """Device bytes this class allocates on ``rank`` for a fit of n_dom + n_bdy collocation points over ``world`` ranks, by buffer: what
        __init__, build(), factor() and solve() hold at their peak (DESIGN.md section 6 prints this table for configs[4]).  Pure arithmetic:
        usable without a GPU."""
        M = 4 * n_dom + n_bdy
        nblk = (M + BLK - 1) // BLK
        Mp = nblk * BLK
        owned = len(owned_blocks(nblk, rank, world))
        blk = BLK * BLK * 8
        cnt = (nblk - 1 + world - 1) // world                 # blocks per rank in the widest column panel (step k = 0)
        out = {"M": M, "block_rows": nblk, "owned_block_rows": owned,
               "panel_R": owned * BLK * Mp * 8,                # build(): this rank's block rows, full width (columns beyond the block's own stay zero)
               # factor(): every diagonal factor, replicated (broadcast in step 1); solve(): -L_GG^-1 of every group of GROUP block rows, replicated
               # (and the transposes of those inverses: both substitution directions are row-major products)
               "diag_factors": nblk * blk + 2 * ((nblk + GROUP - 1) // GROUP) * GROUP * GROUP * blk,
               "collocation_f32": (n_dom + n_bdy) * (d + 1) * 4,
               # factor(), per step: the send buffer, the gathered panel and its reordered copy; with look-ahead the panels of steps k and k + 1 coexist
               # the two panels of a pair side by side (_pair_chain), again in two generations
               "panel_exchange_peak": cnt * blk + world * cnt * blk + 2 * (nblk - 1) * blk + 2 * 2 * (nblk - 2) * blk,
               "vectors": 6 * Mp * 8 + 64 * Mp * 8}            # solve() / matvec(): right-hand side, y, x, accumulator, local rows, output; the 64 row groups' partial sums of the ordered sweep
        out["total"] = sum(v for k, v in out.items() if k not in ("M", "block_rows", "owned_block_rows"))
        return out

    # -------------------------------------------------------------------------------------------- Gram
    def build(self, reuse=None):
        """``reuse``: the panel of a finished factorisation of the same shape, overwritten in place -- a second factorisation then needs no second
        panel and does not depend on the caching allocator handing a 100+ GB block back (at M = 140 002, one 157 GB panel on a 288 GB GPU, the
        freed panel stayed reserved and the second one did not fit)."""
        torch = _lib.require_gpu()
        lib, s = self.lib, _lib.stream_ptr()
        shape = (len(self.mine) * BLK, self.Mp)
        if reuse is not None and tuple(reuse.shape) == shape and reuse.dtype == torch.float64 and reuse.is_contiguous():
            self.R = reuse.zero_()
        else:
            self.R = torch.zeros(shape, dtype=torch.float64, device="cuda")
        for slot, i in enumerate(self.mine):
            row0 = i * BLK
            nrows = max(0, min(BLK, self.M - row0))
            ncols = min((i + 1) * BLK, self.M)
            if self.compat_idx is not None:
                _lib.check(lib.scasml_gp_gram_compat_rows(self.d, self.a, _lib.ptr(self.xd), self.n_dom, _lib.ptr(self.xb), self.n_bdy,
                                                          self.compat_idx.ctypes.data_as(C.c_void_p), self.gram_bits, row0, nrows, ncols,
                                                          self._ptr(self.R, slot * BLK, 0, self.Mp), self.Mp, s), "gp_gram_compat_rows")
            else:
                _lib.check(lib.scasml_gp_gram_rows(self.d, self.a, _lib.ptr(self.xd), self.n_dom, _lib.ptr(self.xb), self.n_bdy, row0, nrows,
                                                   ncols, self._ptr(self.R, slot * BLK, 0, self.Mp), self.Mp, s), "gp_gram_rows")
            blk = self.R[slot * BLK:(slot + 1) * BLK, row0:row0 + BLK]
            dg = blk.diagonal()
            if self.round_diag and nrows > 0:             # float16(K + nugget I): the entries are float16 already, the diagonal moves (:268)
                _lib.check(lib.scasml_round16_diag(self._ptr(self.R, slot * BLK, row0, self.Mp), nrows, self.Mp, self.nugget, s), "round16_diag")
            elif not self.round_diag:
                dg[:nrows] += self.nugget                 # K + nugget I (models/GP.py:260-267)
            dg[nrows:] = 1.0                              # identity padding beyond M
        return self

    # -------------------------------------------------------------------------------------------- factor
    def _panel(self, k):
        """Block column k on the current stream: the owner factors the diagonal block and broadcasts it, every rank solves its blocks of
        the column against it, one all-gather assembles the column panel in global block order.  -> P ((nb - k - 1) * BLK, BLK) or None."""
        torch = _lib.require_gpu()
        lib, s, cm = self.lib, _lib.stream_ptr(), self.comm
        R, Mp, nb, w, rank = self.R, self.Mp, self.nblk, cm.world, cm.rank
        owner = k % w
        Lkk = torch.empty((BLK, BLK), dtype=torch.float64, device="cuda")
        if rank == owner:
            slot = (k - rank) // w
            Lkk.copy_(R[slot * BLK:(slot + 1) * BLK, k * BLK:(k + 1) * BLK])
            _lib.check(lib.scasml_cholesky(_lib.ptr(Lkk), BLK, 0.0, _lib.ptr(self.info), s), "cholesky(diag)")
            self.bad += (self.info != 0).to(torch.float64)                # accumulated on the device: no host read inside the loop
            R[slot * BLK:(slot + 1) * BLK, k * BLK:(k + 1) * BLK] = Lkk
        cm.broadcast(Lkk, owner)
        self.diag[k] = Lkk
        rest = nb - k - 1
        if rest == 0:
            return None
        s0 = self._slot0(k)
        rows = (len(self.mine) - s0) * BLK
        if rows:
            _lib.check(lib.scasml_trsm_right_lt(_lib.ptr(Lkk), BLK, BLK, self._ptr(R, s0 * BLK, k * BLK, Mp), Mp, rows, s), "trsm_right_lt")
        # all-gather the column panel in global block order: rank q = (k + 1 + j) % w holds blocks k + 1 + j, k + 1 + j + w, ...
        cnt = (rest + w - 1) // w
        send = torch.zeros((cnt * BLK, BLK), dtype=torch.float64, device="cuda")
        if rows:
            send[:rows] = R[s0 * BLK:, k * BLK:(k + 1) * BLK]
        got = cm.all_gather(send).view(w, cnt, BLK, BLK)
        order = [(k + 1 + j) % w for j in range(w)]
        return got[order].transpose(0, 1).reshape(cnt * w * BLK, BLK)[:rest * BLK].contiguous()

    def _update(self, k, P, col0, ncols):
        """Trailing update of step k on the current stream, restricted to block columns [col0, col0 + ncols): this rank's block rows
        > k minus (their column-k blocks) x (the panel rows of those columns)^T, tiles above the block diagonal skipped."""
        lib, s, Mp, w = self.lib, _lib.stream_ptr(), self.Mp, self.comm.world
        s0 = self._slot0(k)
        rows = (len(self.mine) - s0) * BLK
        if rows and ncols > 0:
            _lib.check(lib.scasml_gemm_nt_sub(self._ptr(self.R, s0 * BLK, col0 * BLK, Mp), Mp, rows, ncols * BLK,
                                              self._ptr(self.R, s0 * BLK, k * BLK, Mp), Mp,
                                              C.c_void_p(P.data_ptr() + 8 * (col0 - k - 1) * BLK * BLK), BLK, BLK,
                                              self.mine[s0], w, col0, s), "gemm_nt_sub")

    def _update_pair(self, k, P2, col0, ncols):
        """Trailing update by the PAIR of block columns (k, k + 1) on the current stream, block columns [col0, col0 + ncols): ONE product with
        K = 2 * BLK -- this rank's block rows > k + 1, their column blocks k and k + 1 (adjacent in R), against the two gathered panels side by
        side (P2: rows of blocks k + 2 .., 2 * BLK columns).  The 128 x 128 tile's read-modify-write of C is paid once per 512 columns of K
        (scasml_gemm_nt_sub on a C of 16 384^2: 49.5 TFLOP/s at K = 256, 59 at K = 512; profiles/r06_f64_update_tile.txt)."""
        lib, s, Mp, w = self.lib, _lib.stream_ptr(), self.Mp, self.comm.world
        s0 = self._slot0(k + 1)
        rows = (len(self.mine) - s0) * BLK
        if rows and ncols > 0:
            _lib.check(lib.scasml_gemm_nt_sub(self._ptr(self.R, s0 * BLK, col0 * BLK, Mp), Mp, rows, ncols * BLK,
                                              self._ptr(self.R, s0 * BLK, k * BLK, Mp), Mp,
                                              C.c_void_p(P2.data_ptr() + 8 * (col0 - k - 2) * BLK * 2 * BLK), 2 * BLK, 2 * BLK,
                                              self.mine[s0], w, col0, s), "gemm_nt_sub")

    def _pair_chain(self, k):
        """Block columns k and k + 1 on the current stream: panel k, its update of block column k + 1 alone, panel k + 1.  -> the two panels
        side by side for the block rows > k + 1, or None when nothing lies beyond them."""
        torch = _lib.require_gpu()
        Pk = self._panel(k)
        if Pk is None:
            return None
        self._update(k, Pk, k + 1, 1)
        Pk1 = self._panel(k + 1)
        if Pk1 is None:
            return None
        return torch.cat([Pk[BLK:], Pk1], dim=1)

    def factor(self, lookahead=True, pair=True):
        """Right-looking blocked Cholesky over the block rows.  ``pair`` (default): block columns are eliminated two at a time and the
        trailing matrix is updated once per pair with K = 512 (_update_pair); the same collectives as one column at a time, half the
        trailing launches.  With ``lookahead`` the chain of the next pair -- diagonal factors, broadcasts, panel solves, all-gathers, the one
        narrow update between its two columns: short dependent kernels and latency-bound collectives -- runs on a second stream while the
        caller's stream applies the current pair to the block columns beyond the next one (the single-GPU factorisation gained 8 % from the
        same reordering).  The only orderings needed: the chain of the next pair follows the update of ITS block columns by
        the current pair, and every update by the next pair follows that chain.  lookahead=False issues the same operations on one stream
        (bit-identical results); pair=False is the one-column-at-a-time sequence of round 5."""
        torch = _lib.require_gpu()
        cm, nb = self.comm, self.nblk
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream() if lookahead else main
        step = 2 if pair else 1
        chain = self._pair_chain if pair else self._panel
        update = self._update_pair if pair else self._update
        P = chain(0)
        ready = torch.cuda.Event()
        ready.record(main)
        k = 0
        while P is not None:
            main.wait_event(ready)                                         # the panels of this step are complete
            first = k + step                                               # first block column beyond this step
            nxt = min(step, nb - first)
            update(k, P, first, nxt)                                       # the next step's own block columns first: its chain needs only these
            done = torch.cuda.Event()
            done.record(main)
            side.wait_event(done)
            with torch.cuda.stream(side):
                Pn = chain(first)
                ready = torch.cuda.Event()
                ready.record(side)
            update(k, P, first + nxt, nb - first - nxt)                    # the rest of the trailing matrix, under the next chain
            if Pn is not None:
                Pn.record_stream(main)
            P.record_stream(side)
            P = Pn
            k = first
        main.wait_event(ready)
        # a failed pivot is seen by the block's owner only: every rank learns of it through ONE all-reduce after the loop, so that all
        # ranks raise together instead of the others walking into the collectives of solve() without the one that raised
        failed = cm.all_reduce(self.bad.clone())
        if float(failed.item()) != 0.0:
            raise ValueError("distributed Cholesky: K + nugget I is not positive definite (%d diagonal block(s) failed)" % int(failed.item()))
        return self

    # -------------------------------------------------------------------------------------------- substitutions
    def _local_rows(self, v):
        """The owned block rows of a replicated vector of length Mp, stacked."""
        return v.view(self.nblk, BLK)[self._mine_idx()].reshape(-1).clone() if self.mine else v.new_zeros(0)

    def _mine_idx(self):
        torch = _lib.require_gpu()
        if self._mine_dev is None:
            self._mine_dev = torch.as_tensor(self.mine, device="cuda", dtype=torch.long)
        return self._mine_dev

    def _groups(self):
        return [(g0, min(g0 + GROUP, self.nblk)) for g0 in range(0, self.nblk, GROUP)]

    def _neg_inverse_groups(self):
        """-L_GG^-1 of every diagonal SUPER-block (GROUP block rows = 1024 rows), replicated, made once after factor(): each block row of the
        super-block comes from its owner through one all-reduce per group (zeros from everyone else), then one blocked triangular solve against -I.
        A substitution step over a whole group is then a single 1024 x 1024 product: the chains are launch- and collective-latency bound
        (M = 70 001, one rank: 548 collectives and ~2 200 launches per solve with per-block steps)."""
        torch = _lib.require_gpu()
        if self.ninv is None:
            lib, s, cm = self.lib, _lib.stream_ptr(), self.comm
            w, rank = cm.world, cm.rank
            out = []
            for g0, g1 in self._groups():
                n = (g1 - g0) * BLK
                LG = torch.zeros((n, n), dtype=torch.float64, device="cuda")
                for k in range(g0, g1):
                    if k % w == rank:
                        slot = (k - rank) // w
                        LG[(k - g0) * BLK:(k - g0 + 1) * BLK, :(k - g0 + 1) * BLK] = self.R[slot * BLK:(slot + 1) * BLK, g0 * BLK:(k + 1) * BLK]
                cm.all_reduce(LG)
                LG = torch.tril(LG).contiguous()
                NI = torch.zeros((n, n), dtype=torch.float64, device="cuda")
                NI.diagonal().fill_(-1.0)
                _lib.check(lib.scasml_trsm_lower(_lib.ptr(LG), n, _lib.ptr(NI), n, 0, s), "trsm(group^-1)")
                out.append((NI, NI.T.contiguous()))     # and its transpose: the backward step is then a row-major product too (one wave per row)
            self.ninv = out
        return self.ninv

    def solve(self, b):
        """x = (L L^T)^-1 b for a replicated b (length M; a CUDA float64 tensor); returns x replicated.  Bitwise reproducible between runs."""
        torch = _lib.require_gpu()
        lib, s, cm = self.lib, _lib.stream_ptr(), self.comm
        R, Mp, w, rank = self.R, self.Mp, cm.world, cm.rank
        ninv, groups = self._neg_inverse_groups(), self._groups()
        bp = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        bp[:self.M] = b
        loc = self._local_rows(bp)
        rhs = torch.zeros(Mp, dtype=torch.float64, device="cuda")   # group G: the up-to-date right-hand side of its step, replicated by the all-reduce
        y = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        for gi, (g0, g1) in enumerate(groups):                 # forward: L y = b, right-looking, a group of block rows per step
            n = (g1 - g0) * BLK
            seg, yG = rhs[g0 * BLK:g1 * BLK], y[g0 * BLK:g1 * BLK]
            for k in range(g0, g1):                            # the owners' pieces (zeros elsewhere: the sum is a copy)
                if k % w == rank:
                    slot = (k - rank) // w
                    seg[(k - g0) * BLK:(k - g0 + 1) * BLK].copy_(loc[slot * BLK:(slot + 1) * BLK])
            cm.all_reduce(seg)
            _lib.check(lib.scasml_gemv_sub(_lib.ptr(ninv[gi][0]), n, n, n, _lib.ptr(seg), _lib.ptr(yG), 0, s), "gemv(group^-1)")   # y_G = L_GG^-1 b_G
            s0 = self._slot0(g1 - 1)
            rows = (len(self.mine) - s0) * BLK
            if rows:
                _lib.check(lib.scasml_gemv_sub(self._ptr(R, s0 * BLK, g0 * BLK, Mp), Mp, rows, n, _lib.ptr(yG),
                                               C.c_void_p(loc.data_ptr() + 8 * s0 * BLK), 0, s), "gemv_sub")
        x = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        acc = torch.zeros(Mp, dtype=torch.float64, device="cuda")   # this rank's share of -sum_{i > G} L_iG^T x_i, all groups
        for gi in range(len(groups) - 1, -1, -1):              # backward: L^T x = y
            g0, g1 = groups[gi]
            n = (g1 - g0) * BLK
            tG, xG = rhs[g0 * BLK:g1 * BLK], x[g0 * BLK:g1 * BLK]
            tG.copy_(acc[g0 * BLK:g1 * BLK])
            cm.all_reduce(tG)
            tG.add_(y[g0 * BLK:g1 * BLK])
            _lib.check(lib.scasml_gemv_sub(_lib.ptr(ninv[gi][1]), n, n, n, _lib.ptr(tG), _lib.ptr(xG), 0, s), "gemv(group^-T)")   # x_G = L_GG^-T t_G
            if g0 > 0:                                         # the group's owners fold x_G into their accumulators over the columns before the group
                mine_in = [k for k in range(g0, g1) if k % w == rank]
                if w == 1:                                     # one rank: the group's block rows are one contiguous piece of the stack
                    _lib.check(lib.scasml_gemv_sub(self._ptr(R, g0 * BLK, 0, Mp), Mp, n, g0 * BLK, _lib.ptr(xG), _lib.ptr(acc), 1, s), "gemv_sub^T")
                else:
                    for k in mine_in:
                        slot = (k - rank) // w
                        _lib.check(lib.scasml_gemv_sub(self._ptr(R, slot * BLK, 0, Mp), Mp, BLK, g0 * BLK, _lib.ptr(x[k * BLK:(k + 1) * BLK]),
                                                       _lib.ptr(acc), 1, s), "gemv_sub^T")
        return x[:self.M].clone()

    def matvec(self, v):
        """K_p v = L (L^T v) for a replicated v (length M): TWO sweeps over the rank's stacked panel, one launch each, and two collectives of Mp
        doubles.  The sweeps know the block map (scasml_gemv_sub_tri / scasml_gemv_t_sub_ordered_tri): a block row is read up to its diagonal
        block only -- the triangle's bytes, not the full width's (round 6; before, the zeros of the strict upper part were swept too: twice the
        bytes).  The transposed sweep adds its row groups' partial sums in fixed order: bitwise reproducible."""
        torch = _lib.require_gpu()
        lib, s, cm = self.lib, _lib.stream_ptr(), self.comm
        R, Mp = self.R, self.Mp
        rows = len(self.mine) * BLK
        vp = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        vp[:self.M] = v
        acc = torch.zeros(Mp, dtype=torch.float64, device="cuda")          # -(L^T v), summed over ranks
        if rows:
            need = int(lib.scasml_gemv_t_ordered_scratch(rows, Mp))
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float64, device="cuda")
            _lib.check(lib.scasml_gemv_t_sub_ordered_tri(_lib.ptr(R), Mp, rows, Mp, _lib.ptr(self._local_rows(vp)), _lib.ptr(acc), _lib.ptr(self._scratch),
                                                         self._scratch.numel(), self.mine[0], cm.world, s), "gemv_t_sub_ordered_tri")
        cm.all_reduce(acc)
        out = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        if rows:
            loc = torch.zeros(rows, dtype=torch.float64, device="cuda")    # u_i = -L_i acc = L_i (L^T v), this rank's rows
            _lib.check(lib.scasml_gemv_sub_tri(_lib.ptr(R), Mp, rows, Mp, _lib.ptr(acc), _lib.ptr(loc), self.mine[0], cm.world, s), "gemv_sub_tri")
            out.view(self.nblk, BLK)[self._mine_idx()] = loc.view(-1, BLK)
        cm.all_reduce(out)
        return out[:self.M].clone()

    def gather_factor(self):
        """The full lower factor on every rank (tests at small M only)."""
        torch = _lib.require_gpu()
        L = torch.zeros((self.Mp, self.Mp), dtype=torch.float64, device="cuda")
        for slot, i in enumerate(self.mine):
            L[i * BLK:(i + 1) * BLK, :(i + 1) * BLK] = self.R[slot * BLK:(slot + 1) * BLK, :(i + 1) * BLK]
        self.comm.all_reduce(L)
        return torch.tril(L)[:self.M, :self.M]


class DistributedGP:
    """GPsolver at sizes one GPU cannot hold: distributed factor + matrix-free Newton-CG.  After fit() every rank holds
    right_vector (replicated, M doubles) and can hand it to GP.load_right_vector for the (root-sharded) evaluation."""

    def __init__(self, gp, comm=None):
        """gp.compat == "reference" (the default GP): the fit builds the Gram as coded by the reference (scasml_gp_gram_compat_rows), factors it
        for the Newton iteration, then -- as models/GP.py:268, 599, 719 -- rounds z4 and the diagonal of K + nugget I to float16 and solves the
        rounded matrix for right_vector with a second distributed factorisation: the estimator of every other configuration.
        gp.compat is None: the documented operators (scasml_gp_gram_rows)."""
        self.gp = gp
        self.comm = comm or Comm()
        self.cg_iterations = []
        self.gauss_newton_steps = 0

    def fit(self, x_t_domain, x_t_boundary, GN_steps=20, cg_tol=1e-10, cg_max=400, progress=None):
        """cg_tol: relative residual at which the inner conjugate-gradient solve of a Newton step stops.  1e-10 (default) reproduces the dense
        Newton iterates of the single-GPU path step for step (right_vector to 2.6e-14, identical float16 predictions at M = 70 001), and every
        product is bitwise reproducible between runs (substitutions and K_p v add in fixed order).  "adaptive" is the inexact-Newton forcing term
        min(1e-2, |grad_k| / |grad_0|) (Eisenstat-Walker): early steps are solved loosely, the last ones tightly -- the same loss to six digits and
        the same stopping rule (models/GP.py:521: |grad| < 1e-5) in about half of the products.  Its right_vector is NOT the dense path's to
        the last digits: measured 3.2e-3 (relative, max norm) at M = 2940 and at M = 70 001 alike, predictions up to one float16 ulp (4.9e-4) off.
        The cause is the surrogate as coded, not the minimisation: z4 = float16(time_der_rep(sol)) (models/GP.py:719), so a sol that differs in its
        eighth digit rounds a few of the N entries of z4 the other way, and K_p^-1 amplifies those float16 ulps; one more tightly solved Newton step
        does not remove it (measured: 5.5e-3 after it).  Only iterates that follow the dense path to 1e-14 -- the default tolerance -- round alike.
        Opt-in, for fits where a right_vector within the reference's own rounding noise is acceptable (tests/test_gpu_dist_gp.py bounds it).
        progress: optional callable(str), told about every Newton step (long fits on a shared box must show signs of life)."""
        torch = _lib.require_gpu()
        lib, s = _lib.load(), _lib.stream_ptr()
        gp = self.gp
        eq_id, d, sig, mu = int(gp.equation.eq_id), int(gp.d), float(gp.equation.sigma()), float(gp.equation.mu())
        compat_idx = gp.laplacian_idx if getattr(gp, "compat", None) == "reference" else None
        xd16, xb16 = np.asarray(x_t_domain, dtype=np.float32), np.asarray(x_t_boundary, dtype=np.float32)
        graph = bool(getattr(gp, "f16_graph", False)) and compat_idx is not None and \
            np.array_equal(xd16.astype(np.float16).astype(np.float32), xd16) and np.array_equal(xb16.astype(np.float16).astype(np.float32), xb16)
        ch = DistCholesky(d, 1.0 / float(gp.sigma) ** 2, x_t_domain, x_t_boundary, gp.nugget, self.comm, compat_idx=compat_idx, f16_graph=graph,
                          f16_extra=getattr(gp, "_f16_extra", 0)).build().factor()
        self.chol = ch
        N, Nb, M = ch.n_dom, ch.n_bdy, ch.M
        bdy_g = torch.as_tensor(np.asarray(gp.bdy_g(np.asarray(x_t_boundary)), dtype=np.float64), device="cuda").contiguous()
        sol = torch.zeros(3 * N, dtype=torch.float64, device="cuda")
        b = torch.empty(M, dtype=torch.float64, device="cuda")
        damping = 1e-4                                              # models/GP.py:490

        def residual(sol_):
            _lib.check(lib.scasml_gp_newton_b(eq_id, d, sig, mu, _lib.ptr(sol_), _lib.ptr(bdy_g), N, Nb, _lib.ptr(b), s), "gp_newton_b")
            Ab = ch.solve(b)
            return float(torch.dot(b, Ab)), Ab

        def jtv(w, Ab=None, v=None):
            out = torch.empty(3 * N, dtype=torch.float64, device="cuda")
            _lib.check(lib.scasml_gp_newton_jtv(eq_id, d, sig, mu, _lib.ptr(sol), _lib.ptr(w), _lib.ptr(Ab), _lib.ptr(v), 2.0, N, Nb,
                                                _lib.ptr(out), s), "gp_newton_jtv")
            return out

        def hess(v, Ab):
            jv = torch.empty(M, dtype=torch.float64, device="cuda")
            _lib.check(lib.scasml_gp_newton_jv(eq_id, d, sig, mu, _lib.ptr(sol), _lib.ptr(v), N, Nb, _lib.ptr(jv), s), "gp_newton_jv")
            return jtv(ch.solve(jv), Ab, v if Ab is not None else None) + damping * v

        rows = torch.cat([torch.arange(0, N), torch.arange(N + Nb, 2 * N + Nb), torch.arange(3 * N + Nb, M)]).cuda()   # z1, z3, z5 rows of b

        def precond(r):
            """P r = 1/2 S K_p S^T r."""
            v = torch.zeros(M, dtype=torch.float64, device="cuda")
            v[rows] = r
            return 0.5 * ch.matvec(v)[rows]

        def cg(rhs, Ab, tol):
            """(H + damping I) x = rhs by preconditioned conjugate gradients; None on negative curvature."""
            x = torch.zeros_like(rhs)
            r = rhs.clone()
            z = precond(r)
            p = z.clone()
            rz = float(torch.dot(r, z))
            stop = tol * float(torch.linalg.vector_norm(rhs))
            it = 0
            while float(torch.linalg.vector_norm(r)) > stop and it < cg_max:
                hp = hess(p, Ab)
                php = float(torch.dot(p, hp))
                if not php > 0.0:
                    return None, it
                alpha = rz / php
                x.add_(p, alpha=alpha)
                r.add_(hp, alpha=-alpha)
                z = precond(r)
                rz_new = float(torch.dot(r, z))
                p.mul_(rz_new / rz).add_(z)
                rz = rz_new
                it += 1
            return x, it

        loss, Ab = residual(sol)
        hist, self.grad_norms = [loss], []
        for _ in range(GN_steps):                                       # models/GP.py:515-588
            grad = jtv(Ab)
            self.grad_norms.append(float(torch.linalg.vector_norm(grad)))
            if self.grad_norms[-1] < 1e-5:                              # :521
                break
            tol = min(1e-2, max(1e-10, self.grad_norms[-1] / self.grad_norms[0])) if cg_tol == "adaptive" else float(cg_tol)
            step, it = cg(-grad, Ab, tol)
            if step is None:                                            # indefinite Hessian: Gauss-Newton operator instead
                self.gauss_newton_steps += 1
                step, it2 = cg(-grad, None, tol)
                it += it2
            self.cg_iterations.append(it)
            sol = sol + step                                            # alpha = 1, :541,573
            loss, Ab = residual(sol)
            hist.append(loss)
            if progress is not None:
                progress("Newton step %d: loss %.6g, %d CG products" % (len(hist) - 1, loss, it))
        gp.loss_history = hist
        gp.grad_norms = self.grad_norms
        gp._sol = sol
        rv = Ab                                                         # right_vector = K_p^-1 z at the final sol (:593-600)
        if compat_idx is not None:
            # z4 = time_der_rep(sol).astype(float16) (:719); right_vector = solve(float16(K_p), z) (:268, 599): a second factorisation, of the
            # matrix with the float16-rounded diagonal, in the memory of the first
            _lib.check(lib.scasml_round16(C.c_void_p(b.data_ptr() + 8 * (2 * N + Nb)), N, s), "round16")
            panel, ch.R = ch.R, None
            ch.diag, ch.ninv, ch._scratch = [None] * ch.nblk, None, None
            torch.cuda.empty_cache()
            ch2 = DistCholesky(d, 1.0 / float(gp.sigma) ** 2, x_t_domain, x_t_boundary, gp.nugget, self.comm, compat_idx=compat_idx,
                               round_diag=True, f16_graph=graph, f16_extra=getattr(gp, "_f16_extra", 0)).build(reuse=panel).factor()
            del panel
            rv = ch2.solve(b)
            self.chol = ch2
        gp.N_domain, gp.N_boundary, gp.phi_dim = N, Nb, M
        gp.load_right_vector(x_t_domain, x_t_boundary, rv.cpu().numpy())
        return gp
