"""CPU-side checks of the product: the C-ABI library loads and exports every symbol
include/scasml_hip.h declares (no compute without a GPU), struct layouts agree, argument errors
come back as codes + messages, the host tables / schedule match the oracle, and the solvers fail
loudly instead of falling back when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from scasml_gp_amd import _lib, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from scasml_gp_amd import _build
    _build.build_library()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "scasml_hip.h")).read()
    declared = set(re.findall(r"\b(scasml_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)


def test_struct_layouts_and_version(lib):
    assert lib.scasml_abi_version() == _lib.ABI_VERSION == 7
    for which, st in enumerate((_lib.Problem, _lib.Rng, _lib.Term, _lib.Plan, _lib.GpModel)):
        assert lib.scasml_sizeof(which) == C.sizeof(st)
    assert C.sizeof(_lib.Plan) < 3900          # travels by value in the kernarg segment (4 KiB)
    assert lib.scasml_point_stride(100) == 112 and lib.scasml_point_stride(20) == 32 and lib.scasml_point_stride(7) == 16


def test_argument_errors_are_codes_not_crashes(lib):
    plan = tables.build_plan("quad", 2, 2, 0.5, True)
    prob = _lib.Problem(300, 0, 0.5, -0.1, 0.25, 1.0)            # d too large
    rc = lib.scasml_picard_tree(C.byref(prob), C.byref(plan), 0, C.c_void_p(8), 4, 0, _lib.Rng(0, 0, 0, 0, 1, 0, 0),
                                None, None, C.c_void_p(8), None, None)
    assert rc == -2 and b"d=300" in lib.scasml_last_error()
    prob.d = 20
    rc = lib.scasml_picard_tree(C.byref(prob), C.byref(plan), 0, C.c_void_p(8), 4, 0, _lib.Rng(0, 0, 0, 2, 2, 0, 0),
                                None, None, C.c_void_p(8), None, None)
    assert rc == -1 and b"rank" in lib.scasml_last_error()
    assert lib.scasml_picard_tree(C.byref(prob), C.byref(plan), 0, None, 0, 0, _lib.Rng(0, 0, 0, 0, 1, 0, 0), None, None, None, None, None) == 0
    assert lib.scasml_points_per_root(C.byref(plan)) == 29
    assert lib.scasml_trsm_lower(C.c_void_p(8), 33, C.c_void_p(8), 1, 0, None) == -2


@pytest.mark.parametrize("variant,n,par", [("quad", 1, 1), ("quad", 2, 2), ("quad", 3, 3), ("quad", 4, 4), ("quad", 2, 4),
                                            ("fh", 2, 3), ("fh", 3, 3), ("fh", 4, 3), ("fh", 3, 2)])
def test_schedule_matches_oracle_counts(variant, n, par):
    from oracle.mlp import reference_counts, site_count
    from oracle.tables import approx_parameters
    tab = approx_parameters(par) if variant == "quad" else None
    plan = tables.build_plan(variant, n, par, 0.5, True)
    assert tables.executed_path_steps(plan) == site_count(variant, n, par, tab)
    assert tables.reference_path_steps(variant, n, par) == reference_counts(variant, n, par, tab)["path_steps"]
    for np_ in range(1, n + 1):
        assert plan.sites[np_] == site_count(variant, np_, par, tab)


@pytest.mark.parametrize("rho", [1, 2, 3, 4, 5])
def test_host_tables_equal_oracle_tables(rho):
    from oracle.tables import approx_parameters
    for a, b in zip(tables.approx_parameters(rho), approx_parameters(rho)):
        assert np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def test_library_normal_table_equals_the_oracles_restatement(lib):
    """The normal transform's table is a committed constant of the library (csrc/normal_table.inc); the oracle restates its
    definition (oracle/philox.py normal_table).  Bit for bit, no GPU needed."""
    from oracle import philox
    rows = lib.scasml_normal_table_rows()
    assert rows == 768
    got = np.empty((rows, 4), dtype=np.float32)
    assert lib.scasml_normal_table(got.ctypes.data_as(C.c_void_p), rows) == 0
    assert np.array_equal(got.view(np.uint32), philox.normal_table().view(np.uint32))
    assert lib.scasml_normal_table(None, rows) == -1 and b"null" in lib.scasml_last_error()
    # a buffer that is too small is refused, not overrun (ABI 2 copied the whole table into whatever it was given)
    small = np.full((rows, 4), 7.0, dtype=np.float32)
    assert lib.scasml_normal_table(small.ctypes.data_as(C.c_void_p), rows - 1) == -1 and b"rows" in lib.scasml_last_error()
    assert (small == 7.0).all()


def test_stale_delta_t_schedule():
    """MLP.py:249 reuses the previous delta_t for the '+' term; ScaSML.py:253 recomputes it."""
    mlp = tables.build_plan("quad", 3, 3, 0.5, True)
    sca = tables.build_plan("quad", 3, 3, 0.5, False)
    t0, t1, t2 = (mlp.term[3][l] for l in range(3))
    assert list(t0.dplus)[:t0.q] == [1.0] * t0.q                       # level 0 never updates delta_t
    assert t1.dplus[0] == 1.0 and t1.dplus[1] == t1.cfrac[0] and t1.dplus[2] == t1.cfrac[1]
    assert t2.dplus[0] == t1.cfrac[t1.q - 1]                           # carried over from the previous level
    for l in range(3):
        s = sca.term[3][l]
        assert list(s.dplus)[:s.q] == list(s.cfrac)[:s.q]


def test_reference_evaluation_counter_formula():
    # MLP n=rho=1: root call adds Mg=1 and, at l=0, per node k (q=2): child call (Mg[0,0]=1) + MC_f=1
    assert tables.reference_evaluation_count("quad", 1, 1, False) == 1 + 2 * (1 + 1)
    # ScaSML adds 1 per f and per g call on top (ScaSML.py:41,59)
    assert tables.reference_evaluation_count("quad", 1, 1, True) == (1 + 1) + 2 * ((1 + 1) + 1 + 1)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    from scasml_gp_amd.solvers.MLP import MLP
    eq = Grad_Dependent_Nonlinear(11)
    with pytest.raises(_lib.ScasmlError):
        MLP(eq).u_solve(1, 1, np.zeros((2, 11), dtype=np.float16))
    with pytest.raises(_lib.ScasmlError):
        GP_Grad_Dependent_Nonlinear(eq, compat=None).GPsolver(np.zeros((4, 11)), np.zeros((2, 11)))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scasml_gp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_abstract_equation_surface_raises_not_implemented():
    """equations/equations.py:15-230: every abstract method exists and raises NotImplementedError (not AttributeError)."""
    from scasml_gp_amd.equations.equations import Equation
    e = Equation(5)
    assert e.n_input == 5 and e.n_output == 1
    for name, nargs in (("PDE_loss", 3), ("gPDE_loss", 2), ("terminal_constraint", 1), ("initial_constraint", 1), ("Dirichlet_boundary_constraint", 1),
                        ("Neumann_boundary_constraint", 1), ("mu", 1), ("sigma", 1), ("f", 3), ("g", 1), ("exact_solution", 1), ("data_loss", 1),
                        ("geometry", 0), ("test_geometry", 0), ("generate_data", 0), ("generate_test_data", 0)):
        with pytest.raises(NotImplementedError):
            getattr(e, name)(*([0] * nargs))


def test_equation_surface_and_sampler():
    from oracle.equation import GradDependentNonlinear
    from scasml_gp_amd.equations.equations import Equation, Grad_Dependent_Nonlinear
    eq = Grad_Dependent_Nonlinear(21)
    assert isinstance(eq, Equation) and eq.uncertainty == 0.1 and eq.norm_estimation == 1
    np.random.seed(3)
    dom, bdy = eq.generate_data(50, 20)
    assert dom.dtype == np.float16 and dom.shape == (50, 21) and bdy.shape == (20, 21)
    assert (np.abs(bdy[:, :-1]).max(axis=1) == 0.5).all() and (bdy[:, -1] >= 0).all() and (bdy[:, -1] <= 0.5).all()
    ora = GradDependentNonlinear(21)
    d64, b64 = dom.astype(np.float64), bdy.astype(np.float64)
    assert np.allclose(eq.exact_solution(d64), ora.exact_solution(d64)) and np.allclose(eq.g(b64), ora.g(b64))
    # float16 rows in, float16 values out -- the reference's own float16 graph (equations/equations.py:259-261, 317-323)
    from oracle.equation import logistic_wave_f16
    assert eq.exact_solution(dom).dtype == np.float16 and np.array_equal(eq.exact_solution(dom), logistic_wave_f16(dom))
    assert np.array_equal(eq.g(bdy), logistic_wave_f16(bdy)) and np.abs(eq.g(bdy).astype(np.float64) - ora.g(b64)).max() < 2e-3
    u, z = np.random.rand(50, 1), np.random.rand(50, 20)
    assert np.allclose(eq.f(dom, u, z), ora.f(dom, u, z)) and eq.mu() == ora.mu()
    with pytest.raises(NotImplementedError):
        Equation(3).f(None, None, None)


@pytest.mark.parametrize("dealt", [False, True])
@pytest.mark.parametrize("variant,n,par,world", [("quad", 3, 3, 3), ("quad", 2, 2, 2), ("fh", 3, 3, 8), ("quad", 3, 3, 1)])
def test_site_kinds_partition_the_tree_under_sample_sharding(lib, variant, n, par, world, dealt):
    """scasml_plan_site_kinds: 1 = u_hat-only site, 0 = Euler-Maruyama site, 2 = owned by another rank; over the
    ranks every site is owned exactly once and the root row by everyone -- round-robin (owner table NULL) or dealt by
    cost (scasml_plan_deal_units)."""
    from oracle.mlp import site_count
    from oracle.tables import approx_parameters
    from scasml_gp_amd.solvers._picard import deal_units
    plan = tables.build_plan(variant, n, par, 0.5, False)
    ppr = lib.scasml_points_per_root(C.byref(plan))
    owners = np.zeros(ppr, dtype=int)
    base = None
    table = deal_units(plan, world)[0] if dealt else None
    tptr = table.ctypes.data_as(C.c_void_p) if dealt else None
    for r in range(world):
        k = np.zeros(ppr, dtype=np.uint8)
        assert lib.scasml_plan_site_kinds(C.byref(plan), r, world, tptr, k.ctypes.data_as(C.c_void_p)) == 0
        owners += k != 2
        if world == 1:
            base = k
    # every site is owned at least once; the only sites two ranks may both evaluate are the nodes whose "+" and "-" addends went to different ranks
    assert (owners[:-1] >= 1).all() and (owners[:-1] <= 2).all() and owners[-1] == world
    if world > 1:
        assert (owners[:-1] == 2).sum() <= sum(int(plan.term[n][l].mc) * int(plan.term[n][l].q) for l in range(1, n))
    if base is not None:
        tab = approx_parameters(par) if variant == "quad" else None
        assert ppr == site_count(variant, n, par, tab) + 1 and base[-1] == 1 and set(base) <= {0, 1, 3, 4}
        assert (base == 1).sum() == 1                              # the root row alone is kind 1; terminal samples are kind 3
        if variant == "quad" and n == 3 and par == 3:              # 234 terminal jumps executed, 431 Euler-Maruyama steps
            assert (base == 3).sum() == 234 and (base == 0).sum() == 380 and (base == 4).sum() == 51
    assert lib.scasml_plan_site_kinds(C.byref(plan), world, world, None, np.zeros(ppr, dtype=np.uint8).ctypes.data_as(C.c_void_p)) == -1


def test_units_are_dealt_by_cost(lib):
    """scasml_plan_deal_units: every unit gets an owner, the loads add up to the tree's cost plus what sharding adds (a node point evaluated by
    the two owners of its addends, replayed path steps), and longest-first dealing balances unequal units.  The figures quoted in DESIGN.md section 6."""
    from scasml_gp_amd.solvers._picard import deal_units, SITE_COST
    from scasml_gp_amd.parallel import sample_units

    quad = tables.build_plan("quad", 3, 3, 0.5, False)
    whole = deal_units(quad, 1)[1]
    # default costs (ABI <= 6) = Euler-Maruyama sites + 0.6 x terminal sites: 665 sites at n = rho = 3, 234 of them terminal
    assert abs(whole.sum() - (665 - 234 + 0.6 * 234)) < 1e-9
    owner, load = deal_units(quad, 2)
    # units: 27 terminal samples + the addends of the NODES of the sample paths: 5 x 4 nodes at level 0 (one addend each), 3 x 3 at level 1
    # and 2 x 3 at level 2 (a "+" and a "-" addend each: the level-l and the level-(l-1) subtree)
    assert len(owner) == sample_units(quad) == 27 + 20 + 2 * 9 + 2 * 6 == 77 and set(owner) == {0, 1}
    assert 0 <= load.sum() - whole.sum() <= 15 + 1e-9 and load.max() / load.mean() < 1.01     # at most the 15 nodes with two addends are evaluated twice
    for cost in (None, SITE_COST["reference"], SITE_COST["documented"], SITE_COST[None]):
        w1 = deal_units(quad, 1, cost)[1].sum()
        for world in (2, 3, 4, 6, 8):                # the largest addend is 79 sites of 665: eight ranks balance to 1 % (whole paths as units: 3.19)
            ow, lw = deal_units(quad, world, cost)
            assert lw.max() / lw.mean() < (1.012 if cost is not SITE_COST[None] else 1.03) and lw.sum() < 1.05 * w1, (cost, world, lw)   # (replay is charged after the dealing)
            # the zero-cost "-" addends of the nine level-1 nodes (their subtree is uz(0) = 0) stay with the node's "+" addend: no second evaluation
            first = 27 + 20
            assert all(ow[first + 2 * i] == ow[first + 2 * i + 1] for i in range(9))
    # the measured weights of the as-coded surrogate: a level l > 0 site costs 0.62, a terminal site 0.50 of a level-0 site; replay 0.04 per step
    o8, l8 = deal_units(quad, 8, SITE_COST["reference"])
    assert abs(deal_units(quad, 1, SITE_COST["reference"])[1][0] - (380 + 0.62 * 51 + 0.50 * 234)) < 1e-9 and l8.max() / l8.mean() < 1.006
    small = tables.build_plan("quad", 2, 2, 0.5, False)
    assert len(deal_units(small, 2)[0]) == 16 and deal_units(small, 2)[1].max() / deal_units(small, 2)[1].mean() < 1.05
    fh = tables.build_plan("fh", 4, 3, 0.5, False)
    o4, l4 = deal_units(fh, 4)
    assert len(o4) == 81 + 81 + 2 * (27 + 9 + 3) == 240 and l4.max() / l4.mean() < 1.01
    o8, l8 = deal_units(fh, 8)
    assert 1.15 < l8.max() / l8.mean() < 1.25        # three level-3 "+" addends of 15 % of the tree each over eight ranks (whole samples: 1.43)
    l1 = deal_units(fh, 1)[1]
    for world in (1, 2, 3, 8, 255):
        o, l = deal_units(fh, world)
        assert o.max() < world and 0 <= l.sum() - l1.sum() <= 39 + 1e-9      # full history draws a node directly: nothing replayed
    bad = np.array([1.0, 0.0, 0.5, 0.0])
    assert lib.scasml_plan_deal_units(C.byref(fh), 2, bad.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), 240, None) == -1
    assert lib.scasml_plan_deal_units(C.byref(fh), 256, None, o.ctypes.data_as(C.c_void_p), 240, None) == -1
    assert lib.scasml_plan_deal_units(C.byref(fh), 2, None, o.ctypes.data_as(C.c_void_p), 10, None) == -1


def test_bench_chooses_the_sample_ranks_by_predicted_efficiency():
    """bench.py's north-star leg: the number of sample ranks is the largest divisor of the rank count whose PREDICTED strong-scaling efficiency --
    the slowest rank's share of the measured unsharded step plus a fixed cost plus the stated all-reduce cost -- reaches the minimum
    (VERDICT r5 item 1b).  Host logic only: the dealing and the arithmetic of the table."""
    import types
    import bench
    from scasml_gp_amd.solvers._picard import deal_units, SITE_COST
    args = types.SimpleNamespace(allreduce_busbw_gbs=100.0, allreduce_latency_us=40.0, min_sample_efficiency=0.9)
    gp = types.SimpleNamespace(compat="reference", eval_geometry=False)
    quad = tables.build_plan("quad", 3, 3, 0.5, False)
    eng = types.SimpleNamespace(plan=lambda n, par: quad, gp=gp)
    table = bench.sample_split_table(eng, 3, 3, 8, 21.7, 1 << 14, 100, args)
    assert [r["sample_ranks"] for r in table] == [1, 2, 4, 8] and [r["root_groups"] for r in table] == [8, 4, 2, 1]
    assert table[0]["predicted_efficiency"] == 1.0 and table[0]["allreduce_ms_stated"] == 0.0
    # 8 sample ranks: 6.6 MB over 8 ranks at the stated 100 GB/s + 40 us = 0.156 ms; the slowest rank's share of 21.7 ms by the dealt load
    l8, w1 = deal_units(quad, 8, SITE_COST["reference"])[1], deal_units(quad, 1, SITE_COST["reference"])[1][0]
    assert abs(table[3]["allreduce_ms_stated"] - (0.040 + 1.75 * (1 << 14) * 101 * 4 / 100e9 * 1e3)) < 1e-3
    assert abs(table[3]["modelled_rank_ms"] - (21.7 * l8.max() / w1 + bench.SAMPLE_RANK_FIXED_MS)) < 1e-3
    assert abs(table[3]["predicted_efficiency"] - 21.7 / (8 * (table[3]["modelled_rank_ms"] + table[3]["allreduce_ms_stated"]))) < 1e-3
    assert all(table[i]["predicted_efficiency"] >= table[i + 1]["predicted_efficiency"] for i in range(3))      # every further split costs
    assert 0.85 < table[3]["predicted_efficiency"] < 0.95 and table[2]["predicted_efficiency"] > 0.9
    # full history n = 4 on eight ranks: three addends hold 15 % of the tree each -- eight sample ranks are predicted below 0.8, four above 0.9
    fh = tables.build_plan("fh", 4, 3, 0.5, False)
    tfh = bench.sample_split_table(types.SimpleNamespace(plan=lambda n, par: fh, gp=gp), 4, 3, 8, 53.5, 1 << 14, 100, args)
    assert tfh[3]["predicted_efficiency"] < 0.8 < 0.9 < tfh[2]["predicted_efficiency"]
    # a step that is all fixed cost (96 roots): nothing beyond the roots split is predicted to pay
    tiny = bench.sample_split_table(eng, 3, 3, 4, 0.12, 96, 100, args)
    assert max(r["sample_ranks"] for r in tiny if r["predicted_efficiency"] >= 0.9) == 1


@pytest.fixture(scope="module")
def kernel_regs_reports():
    """tools/kernel_regs.py (device-only compile to ISA text, ~50 s per file) for the evaluation kernels and the FP64 linear algebra, run side by
    side once per module."""
    import subprocess, sys, os
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc at %s" % hipcc)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(name):
        return subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_regs.py"), name], capture_output=True, text=True, check=True).stdout
    files = ["gp_eval_bf16.hip", "gp_eval_compat_mfma.hip", "gp_train.hip", "dist_linalg.hip"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        return dict(zip(files, ex.map(run, files)))


def test_no_spills_inside_the_gp_tile_loops(kernel_regs_reports):
    """A register spill or reload inside the tile loop of the GP evaluation kernels is a performance cliff (round 1 measured
    72 ms for a spilling build against 8), and it makes the counted `s_waitcnt vmcnt(N)` of the LDS-DMA hand-over more
    conservative than written.  tools/kernel_regs.py compiles the file to ISA text and reports scratch instructions inside
    loops: only the kernels that drain fully (three LDS slots, `vmcnt(0)`) may have any.  Needs hipcc ($HIPCC or /opt/rocm)."""
    import re
    out = kernel_regs_reports["gp_eval_bf16.hip"]
    assert "gp_eval_bf16_kernel<7, 2, 4, true, 4, true>" in out            # the headline instantiation is there
    for line in out.splitlines():
        if "!!" not in line:
            continue
        ks, split, _, _, _, yexact = re.search(r"<(\d+), (\d+), (\d+), (\w+), (\d+), (\w+)>", line).groups()
        planes = 1 if yexact == "true" else int(split)
        stage_bytes = (planes * int(ks) * 256 + 512) * 4
        bpc = int(re.search(r"<\d+, \d+, \d+, \w+, (\d+),", line).group(1))
        assert 4 * stage_bytes * bpc > 144 * 1024, "spill inside a counted-vmcnt tile loop: " + line


def test_launch_bounds_of_the_as_coded_evaluation_kernel_hold_without_scratch(kernel_regs_reports):
    """gp_eval_compat_mfma.hip sets __launch_bounds__(256, BPC) from a hand-written register estimate; if the estimate is low the
    compiler honours the occupancy bound by spilling, silently.  Pinned from the code object's metadata for every instantiation: the as-coded
    kernel (float16 entries, two planes -- the default and the benchmarked one) uses NO scratch at any d; the opt-in geometry mode may park a few
    dwords (<= 32 bytes, KS = 7..9 and 15 at one plane) but never touches scratch inside a loop."""
    import re
    out = kernel_regs_reports["gp_eval_compat_mfma.hip"]
    seen = 0
    for line in out.splitlines():
        m = re.search(r"gp_eval_compat_mfma_kernel<(\d+), (\d+), (\w+), (\d+)>.*scratch\s+(\d+)\s+vgpr\s+(\d+)", line)
        if not m:
            continue
        seen += 1
        ks, bpc, r16, planes, scratch, vgpr = int(m.group(1)), int(m.group(2)), m.group(3) == "true", int(m.group(4)), int(m.group(5)), int(m.group(6))
        assert "!!" not in line, "scratch traffic inside a loop: " + line
        assert vgpr <= 512 // bpc, line                      # 512 VGPRs per SIMD lane, one 256-thread workgroup = one wave per SIMD
        assert scratch == 0 if r16 else scratch <= 32, line
    assert seen == 3 * 16                                     # KS = 1..16 (d <= 252) x {as coded, geometry with 1 plane, geometry with 2}


def test_fp64_dma_tiles_fit_a_1024_thread_workgroup_without_scratch(kernel_regs_reports):
    """The 128 x 128 FP64 update tile with LDS-DMA operand staging (csrc/f64_tile_dma.hpp) runs as ONE 1024-thread workgroup per CU, four waves per
    SIMD: 128 VGPRs per wave at most, and its counted `s_waitcnt vmcnt(N)` hand-over assumes that nothing else -- no spill, no reload -- issues
    vector-memory operations inside the stage loop (a round-6 draft that streamed several tiles per workgroup spilled with `vmcnt(0)` waits in the
    middle of the output tile's loads).  Pinned from the code objects for the three kernels built on it."""
    import re
    want = {"gp_train.hip": ["chol_update_dma_kernel", "trsm_update_dma_kernel<0>", "trsm_update_dma_kernel<1>"], "dist_linalg.hip": ["gemm_nt_sub_dma_kernel"]}
    for f, names in want.items():
        out = kernel_regs_reports[f]
        for name in names:
            lines = [l for l in out.splitlines() if name in l]
            assert len(lines) == 1, (name, lines)
            m = re.search(r"scratch\s+(\d+)\s+vgpr\s+(\d+)", lines[0])
            assert m and int(m.group(1)) == 0 and int(m.group(2)) <= 128 and "!!" not in lines[0], lines[0]


def test_root_bound_is_cached_for_the_callers_tensor_only():
    """the bound of a temporary (a CPU / non-float32 / non-contiguous input converted by _as_device) must not be served from the
    cache -- the allocator recycles its address for the next call's temporary.  Host logic only: no GPU needed."""
    import torch
    from scasml_gp_amd.solvers._picard import PicardEngine
    eng = PicardEngine.__new__(PicardEngine)
    eng._bound_cache = None
    small, large = torch.full((8, 4), 0.25), torch.full((8, 4), 40.0)
    assert eng._root_bound(small, None, False) == 0.25 and eng._bound_cache is None
    assert eng._root_bound(large, None, False) == 40.0                       # same shape, no cache: reduced again
    own = torch.full((8, 4), 0.5)
    assert eng._root_bound(own, None, True) == 0.5 and eng._bound_cache is not None
    view = own[:4]
    assert eng._root_bound(view, None, True) == 0.5                           # a view: its own key
    own.mul_(6.0)                                                             # in-place change: the version moves, the cache misses
    assert eng._root_bound(own, None, True) == 3.0
    other = torch.full((8, 4), 9.0)
    assert eng._root_bound(other, None, True) == 9.0                          # another tensor object
    del other
    assert eng._bound_cache[0]() is None                                      # held weakly: the cache keeps no tensor alive
    assert eng._root_bound(own, 7.5, True) == 7.5                             # a host bound wins


@pytest.mark.parametrize("nti,ntj,tri", [(1, 1, 1), (7, 7, 1), (9, 9, 1), (33, 33, 1), (274, 274, 1), (5, 5, 0), (33, 11, 0), (9, 130, 0), (274, 91, 0)])
def test_tile_order_of_the_update_kernels_covers_every_tile_once(lib, nti, ntj, tri):
    """The 1-D grid of the 128 x 128 FP64 update kernels (host_common.hpp: super-tiles of 8 x 4 tiles per XCD, the lower triangle's only
    when tri): every tile of the launch belongs to exactly one workgroup, the 32 consecutive workgroups of an XCD (b, b + 8, ...) lie in one
    super-tile, and workgroups without a tile say so."""
    nb = lib.scasml_tile_order_blocks(nti, ntj, tri)
    assert nb > 0 and nb % (8 * 32) == 0
    ti, tj = C.c_int64(), C.c_int64()
    owner, supers = {}, {}
    for b in range(nb):
        r = lib.scasml_tile_order(b, nti, ntj, tri, C.byref(ti), C.byref(tj))
        assert r in (0, 1)
        if r == 0:
            assert (ti.value, tj.value) == (-1, -1)
            continue
        assert (ti.value, tj.value) not in owner
        owner[(ti.value, tj.value)] = b
        supers.setdefault((b % 8, b // 8 // 32), set()).add((ti.value // 8, tj.value // 4))
    want = {(i, j) for i in range(nti) for j in range(ntj) if not tri or j // 4 <= (i // 8) * 2 + 1}   # tiles above the diagonal inside a diagonal super-tile are dealt and skipped by the kernel
    assert set(owner) == want and {(i, j) for i in range(nti) for j in range(ntj) if not tri or j <= i} <= want
    assert all(len(v) == 1 for v in supers.values())
    assert lib.scasml_tile_order_blocks(4, 5, 1) < 0 and lib.scasml_tile_order(0, 4, 4, 0, None, None) < 0


def test_distributed_gp_memory_budget_for_configs4():
    """DESIGN.md section 6 prints DistCholesky.budget for BASELINE configs[4] (d = 250, 83 333 + 16 667 collocation points, 8 ranks); the numbers
    are pinned here, and tests/test_gpu_xl.py checks the same function against what the class really allocates at M = 70 001."""
    from scasml_gp_amd.dist_gp import DistCholesky
    b = DistCholesky.budget(250, 83333, 16667, 8)
    assert (b["M"], b["block_rows"], b["owned_block_rows"]) == (349999, 1368, 171)
    assert b["panel_R"] == 171 * 256 * 350208 * 8 and round(b["panel_R"] / 1e9, 1) == 122.6
    # the replicated diagonal factors + the inverses of the 342 diagonal super-blocks of four block rows (1024 x 1024 each) the substitutions step through
    # (and their transposes)
    assert b["diag_factors"] == 1368 * 256 * 256 * 8 + 2 * 342 * 1024 * 1024 * 8 and 133e9 < b["total"] < 136e9   # fits a 288 GB MI355X more than twice over
    assert 256e9 < DistCholesky.budget(250, 83333, 16667, 4)["total"] < 259e9                # four GPUs: still fits
    one = DistCholesky.budget(250, 16667, 3333, 1)
    assert one["panel_R"] == 274 * 256 * 70144 * 8 and one["M"] == 70001
    assert sum(DistCholesky.budget(250, 16667, 3333, 2, r)["owned_block_rows"] for r in range(2)) == 274


def test_gp_host_views_of_the_collocation_operator():
    """GP.time_der_rep and GP.DF_domain_without_time (models/GP.py:705-743) against the reference's formulas written out: host NumPy, no GPU."""
    from scasml_gp_amd.equations.equations import Grad_Dependent_Nonlinear
    from scasml_gp_amd.models.GP import GP_Grad_Dependent_Nonlinear
    d, N = 20, 7
    gp = GP_Grad_Dependent_Nonlinear(Grad_Dependent_Nonlinear(d + 1))
    gp.N_domain, gp.N_boundary = N, 3
    sol = np.random.default_rng(0).standard_normal(3 * N)
    z1, z3, z5 = sol[:N], sol[N:2 * N], sol[2 * N:]
    s2 = 0.25 ** 2
    F = -s2 * z1 * z5 + (1 / d + s2 / 2) * z5 - (s2 / 2) * z3                               # :717
    assert np.allclose(gp.time_der_rep(sol, np.zeros(N)), F, rtol=1e-14) and np.allclose(gp.time_der_rep(sol, np.ones(N)), F + 1, rtol=1e-14)
    want = np.hstack([-s2 * np.diag(z5), -(s2 / 2) * np.eye(N), -(s2 * np.diag(z1) - (1 / d + s2 / 2) * np.eye(N))]).astype(np.float16)   # :733-743
    got = gp.DF_domain_without_time(sol)
    assert got.dtype == np.float16 and got.shape == (N, 3 * N) and np.array_equal(got, want)


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/scasml_hip.h must compile as C99 with no extensions."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text('#include "scasml_hip.h"\nint main(void) { return (int)sizeof(scasml_plan) > 0 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                    "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_round2_entry_points_validate_their_arguments_without_a_gpu(lib):
    """Argument errors of the reference-compat, distributed and dealing entry points come back as codes + messages before
    anything is launched (so this runs on the CPU box)."""
    p8 = C.c_void_p(8)
    idx = (C.c_int32 * 5)(1, 1, 2, 3, 4)
    assert lib.scasml_gp_gram_compat(20, 0.8, p8, 10, p8, 2, idx, 1, p8, None) == -1 and b"repeated" in lib.scasml_last_error()
    idx = (C.c_int32 * 5)(0, 1, 2, 3, 20)
    assert lib.scasml_gp_gram_compat(20, 0.8, p8, 10, p8, 2, idx, 1, p8, None) == -1 and b"outside [0, d)" in lib.scasml_last_error()
    idx = (C.c_int32 * 5)(0, 1, 2, 3, 4)
    assert lib.scasml_gp_gram_compat(4, 0.8, p8, 10, p8, 2, idx, 1, p8, None) == -1          # d < 5
    assert lib.scasml_gp_eval_compat(20, 0.8, 0.25, 0.0, 9, p8, 10, 2, 12, p8, idx, 1, p8, 4, 32, p8, None, None) == -2
    assert b"unknown equation id 9" in lib.scasml_last_error()
    assert lib.scasml_gp_eval_compat(20, 0.8, 0.25, 0.0, 0, p8, 10, 2, 11, p8, idx, 1, p8, 4, 32, p8, None, None) == -1   # ldc < N
    assert lib.scasml_gp_eval_compat(20, 0.8, 0.25, 0.0, 0, p8, 10, 2, 12, p8, idx, 1, p8, 0, 32, p8, None, None) == 0    # empty batch
    assert lib.scasml_gemm_nt_sub(p8, 64, 64, 64, p8, 33, p8, 33, 33, 0, 0, 0, None) == -2 and b"multiple of 32" in lib.scasml_last_error()
    assert lib.scasml_gemm_nt_sub(p8, 10, 64, 64, p8, 64, p8, 64, 64, 0, 0, 0, None) == -1                                # ldc < cols
    assert lib.scasml_trsm_right_lt(p8, 256, 250, p8, 256, 10, None) == -2
    assert lib.scasml_gp_gram_rows(20, 0.8, p8, 10, p8, 2, 40, 8, 42, p8, 42, None) == -1                                  # rows beyond M = 42
    assert lib.scasml_gemv_sub(p8, 4, 8, 8, p8, p8, 0, None) == -1                                                         # lda < cols
    assert lib.scasml_gp_newton_b(5, 20, 0.25, 0.0, p8, p8, 10, 2, p8, None) == -2
    prob = _lib.Problem(10, 0, 0.5, 0.0, 0.25, 1.0)
    plan = tables.build_plan("quad", 1, 1, 0.5, True)
    assert lib.scasml_picard_tree(C.byref(prob), C.byref(plan), 0, p8, 4, 3, _lib.Rng(0, 0, 0, 0, 1, 0, 0), None, None, p8, None, None) == -1
    assert b"site_stride" in lib.scasml_last_error()
