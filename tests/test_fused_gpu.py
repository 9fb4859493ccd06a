"""The persistent fused tail kernel (k_sweep_fused) is the same device code as the three sweep kernels:
solving with and without it must give the same bits."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, s in (("turn90", P.batch_turn90(make, batch=40)), ("obstacles", P.batch_three_obstacles(make, batch=24, dtype=A.F64))):
    s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    tm = s.get_timing()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_K"] = s.get_gains()[0]
    out[name + "_it"] = st["iterations_total"]; out[name + "_status"] = st["status"]; out[name + "_cost"] = st["cost"]
    out[name + "_fused"] = np.array([tm["fused_sweeps"]])
np.savez(sys.argv[1], **out)
'''


def _run(tmp_path, tag, env_extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"{tag}.npz")
    env = dict(os.environ, **env_extra)
    subprocess.run([sys.executable, "-c", _SCRIPT % root, out], check=True, env=env, timeout=600)
    return np.load(out)


def test_fused_tail_matches_separate_kernels_bitwise(tmp_path):
    a = _run(tmp_path, "fused", {})
    b = _run(tmp_path, "plain", {"ALTRO_HIP_NO_FUSED_SWEEP": "1"})
    assert a["turn90_fused"][0] > 0 and a["obstacles_fused"][0] > 0   # the persistent kernel really ran
    assert b["turn90_fused"][0] == 0 and b["obstacles_fused"][0] == 0
    for k in a.files:
        if k.endswith("_fused"):
            continue
        assert np.array_equal(a[k], b[k]), k


_SCRIPT_FF = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
s = P.batch_turn90(make, batch=512, seed=P.SEED_BASE + 3)
s.set_record_history(301)
s.solve()
st = s.get_stats()
X, U = s.get_trajectory()
out = {"X": X, "U": U, "K": s.get_gains()[0], "lam": s.get_duals(), "pen": s.get_penalties()}
for f in st.dtype.names:
    out["st_" + f] = st[f]
bad = np.flatnonzero(st["status"] != 0)
out["bad"] = bad
for b in bad[:4]:
    for f in ("cost", "alpha", "gradient", "cost_decrease", "regularization", "violations", "max_penalty", "improvement_ratio"):
        out["h%%d_%%s" %% (b, f)] = s.get_history(int(b), f)
np.savez(sys.argv[1], **out)
'''


def test_stall_fast_forward_is_bit_identical(tmp_path):
    """ALTRO_HIP_FAST_FORWARD_STALLS (opt-in): the straggler instances sit at a fixed point -- a rejected line
    search repeated ~100 times -- and counting those repetitions instead of recomputing them must not change one
    bit of the statistics, the per-iteration history, the trajectories, the gains or the multipliers."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for tag, env in (("plain", {}), ("ff", {"ALTRO_HIP_FAST_FORWARD_STALLS": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_FF % root, out], check=True, env=dict(os.environ, **env), timeout=600)
        res.append(np.load(out))
    a, b = res
    assert len(a["bad"]) > 0  # there are stragglers in this batch
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


_SCRIPT_SW = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, s in (("turn90", P.batch_turn90(make, batch=700)), ("obstacles", P.batch_three_obstacles(make, batch=700, dtype=A.F32))):
    s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_K"] = s.get_gains()[0]; out[name + "_lam"] = s.get_duals()
    out[name + "_it"] = st["iterations_total"]; out[name + "_status"] = st["status"]; out[name + "_cost"] = st["cost"]
np.savez(sys.argv[1], **out)
'''


@pytest.mark.parametrize("switch", [{"ALTRO_HIP_NO_DENSE_EXPANSIONS": "1"}, {"ALTRO_HIP_FWD_SRC": "global"},
                                    {"ALTRO_HIP_FWD_SRC": "lds"}, {"ALTRO_HIP_FWD_PER_WAVE": "1"},
                                    {"ALTRO_HIP_SPECULATION": "off"},
                                    {"ALTRO_HIP_SPECULATION": "free"}, {"ALTRO_HIP_SPECULATION": "wave"},
                                    {"ALTRO_HIP_SPECULATION": "free", "ALTRO_HIP_DEBUG_POISON": "12345678,mix"},
                                    {"ALTRO_HIP_SWEEP_LOOP": "0"}, {"ALTRO_HIP_SWEEP_LOOP": "1", "ALTRO_HIP_LOOP_PER_CU": "1"},
                                    {"ALTRO_HIP_PERSIST_AT": "600"}, {"ALTRO_HIP_CHAINS": "4"}, {"ALTRO_HIP_CHAINS": "3", "ALTRO_HIP_PERSIST_AT": "100"}, {"ALTRO_HIP_DEBUG_POISON": "ffffffff"},
                                    {"ALTRO_HIP_DEBUG_POISON": "12345678,mix"}, {"ALTRO_HIP_BEGIN_SOLVE": "split"}],
                         ids=lambda d: "-".join(f"{k}={v}" for k, v in d.items()))
def test_launch_variants_are_bit_identical(tmp_path, switch):
    """The batched sweeps have several launch variants chosen by measurements (dense / list-addressed expansions with
    the list rebuilt in neighbour order, rollout inputs staged in LDS or read from global memory, instances per
    workgroup, the persistent kernel with and without the speculative backward pass of its fourth wave): none of them may
    change a bit of the result.  The same holds for the point where the persistent kernel takes over, and for the content
    of the LDS and of the candidate buffer before each kernel (ALTRO_HIP_DEBUG_POISON: NaN words, mixed words): no kernel
    may compute with memory the solve has not written."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, env_extra):
        out = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_SW % root, out], check=True, env=dict(os.environ, **env_extra), timeout=600)
        return np.load(out)

    a, b = run("default", {}), run("switch", switch)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


_SCRIPT_CAND = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, s in (("turn90", P.batch_turn90(make, batch=1024, seed=P.SEED_BASE + 3)),
                ("obstacles32", P.batch_three_obstacles(make, batch=768, dtype=A.F32)),
                ("quad12", P.batch_quadrotor12(make, batch=96, dtype=A.F32))):
    s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_lam"] = s.get_duals(); out[name + "_c"] = s.get_constraint_values()
    for f in st.dtype.names:
        out[name + "_st_" + f] = st[f]
np.savez(sys.argv[1], **out)
'''


def test_candidate_layouts_are_bit_identical(tmp_path):
    """CandLayout (altro_kernels.hpp): the batched forward kernel keeps a candidate slot for the first `front` line-search
    trials and the last live one, and REPLAYS a deeper winner.  front = 19 stores every trial (the round-3 layout); 8 is the
    (6 is the default); 2 and 0 replay most / every accepted step -- every layout must produce the same bits (trajectories,
    multipliers, the stale c_ of quirk Q6, every statistic), with and without the persistent tail kernel."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, env_extra):
        out = str(tmp_path / f"cand_{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_CAND % root, out], check=True, env=dict(os.environ, **env_extra), timeout=900)
        return np.load(out)
    ref = run("all", {"ALTRO_HIP_CAND_FRONT": "19", "ALTRO_HIP_NO_FUSED_SWEEP": "1"})
    assert (ref["turn90_st_status"] == 0).mean() > 0.9 and ref["obstacles32_st_iterations_total"].max() > 100
    for tag, env in (("6", {"ALTRO_HIP_NO_FUSED_SWEEP": "1"}), ("8", {"ALTRO_HIP_CAND_FRONT": "8", "ALTRO_HIP_NO_FUSED_SWEEP": "1"}), ("2", {"ALTRO_HIP_CAND_FRONT": "2", "ALTRO_HIP_NO_FUSED_SWEEP": "1"}),
                     ("0", {"ALTRO_HIP_CAND_FRONT": "0", "ALTRO_HIP_NO_FUSED_SWEEP": "1"}), ("default", {}),
                     # ... and with LDS and the candidate buffer poisoned before every kernel: a replayed winner's slot is
                     # really rewritten before phase 2 reads it, an unwritten slot would compute with NaN words
                     ("0_poisoned", {"ALTRO_HIP_CAND_FRONT": "0", "ALTRO_HIP_NO_FUSED_SWEEP": "1", "ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"}),
                     ("6_poisoned", {"ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"})):
        got = run(tag, env)
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (tag, k)


_SCRIPT_TWIN = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
cases = (("turn90_512", lambda: P.batch_turn90(make, batch=512, seed=P.SEED_BASE + 3)),        # persistent kernel from the first sweep, ~12 stragglers
         ("turn90_2304", lambda: P.batch_turn90(make, batch=2304, seed=P.SEED_BASE + 3)),      # chains of sweeps, hand-over, ~55 stragglers
         ("obstacles_192", lambda: P.batch_three_obstacles(make, batch=192, dtype=A.F64)),   # lock-step speculation (circle constraints)
         ("obstacles_r32", lambda: P.batch_three_obstacles(make, batch=192, dtype=A.F32)))   # fp32 records
for name, fac in cases:
    s = fac()
    for rep in range(2):   # (the second solve reuses mailboxes and shadow columns)
        s.reset_trajectory()
        s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    tm = s.get_timing()
    out[name + "_X"] = X; out[name + "_U"] = U
    K, d = s.get_gains()
    out[name + "_K"] = K; out[name + "_d"] = d
    out[name + "_lam"] = s.get_duals(); out[name + "_pen"] = s.get_penalties(); out[name + "_c"] = s.get_constraint_values()
    for f in st.dtype.names:
        out[name + "_st_" + f] = st[f]
    for k in (0, 50, 100):
        e = s.get_expansion(k)
        for key, v in e.items():
            if k < 100 or key in ("lxx", "lx"):   # (the terminal knot has no dynamics and no control blocks)
                out[name + "_exp%%d_%%s" %% (k, key)] = v
    out[name + "_costs"] = s.get_knot_costs()
    out[name + "_twins"] = np.array([tm["twin_workgroups"]]); out[name + "_ms"] = np.array([tm["total_ms"]])
    out[name + "_iters"] = np.array([tm["instance_iterations"], tm["fused_instance_iterations"], tm["sweeps"]])
    s.close()
np.savez(sys.argv[1], **out)
'''


def test_twin_workgroups_are_bit_identical(tmp_path):
    """Twin workgroups of the persistent kernel (TwinCtl, altro_kernels.hpp): the second half of a straggler's rejection
    streak is computed by a second workgroup on a shadow column, beside the first half, and copied back after the
    primary has confirmed the state the twin assumed.  Every iteration is still executed with the inputs the sequential
    order gives it, so NOTHING may differ from a launch without twins (ALTRO_HIP_TWIN=0): trajectories, gains,
    multipliers, penalties, stored constraint values, expansion records, knot costs, every statistic -- on the batch that
    starts in the persistent kernel, behind the chains of sweeps, with circle constraints (lock-step speculation) and
    with fp32 records."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (third run: with the shadow columns, LDS and the candidate buffer full of NaN words before every solve)
    for tag, env_extra in (("twin", {}), ("solo", {"ALTRO_HIP_TWIN": "0"}), ("twin_poisoned", {"ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"})):
        out = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_TWIN % root, out], check=True, env=dict(os.environ, **env_extra), timeout=900)
        res[tag] = np.load(out)
    a, b = res["twin"], res["solo"]
    for k in a.files:
        if not k.endswith(("_twins", "_ms", "_iters")):
            assert np.array_equal(res["twin_poisoned"][k], b[k]), ("poisoned", k)
    for name in ("turn90_512", "turn90_2304", "obstacles_192", "obstacles_r32"):
        assert a[name + "_twins"][0] > 0 and b[name + "_twins"][0] == 0, name
        print(name, "ms with / without twins", a[name + "_ms"][0], b[name + "_ms"][0], "iterations", a[name + "_iters"], b[name + "_iters"])
    for k in a.files:
        if k.endswith(("_twins", "_ms", "_iters")):
            continue
        assert np.array_equal(a[k], b[k]), (k, np.abs(np.asarray(a[k], float) - np.asarray(b[k], float)).max())
    # the twins really took their share: a batch with stragglers finishes sooner, and every iteration was executed
    assert a["turn90_2304_ms"][0] < 0.9 * b["turn90_2304_ms"][0], (a["turn90_2304_ms"], b["turn90_2304_ms"])  # (measured 0.72 - 0.75)
    assert a["turn90_2304_iters"][0] == b["turn90_2304_iters"][0]       # sum of iterations_total
    # (the units executed by the persistent launch, iters[1], are not compared: which sweep hands over to it depends on
    #  when the host sees the counts, with or without twins -- 5 622 or 5 637 of the 32 097 in different runs)


_SCRIPT_SEG = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
cases = (("obstacles_4096_r32", lambda: P.batch_three_obstacles(make, batch=4096, dtype=A.F32), "al"),   # BASELINE configs[3]: a quarter of the batch ends in 100-iteration streaks
         ("obstacles_2048_f64", lambda: P.batch_three_obstacles(make, batch=2048, dtype=A.F64), "al"),   # fp64 records, two chains of sweeps
         ("obstacles_1024_ilqr", lambda: P.batch_three_obstacles(make, batch=1024, dtype=A.F64), "ilqr"),  # one chain, plain iLQR on the penalised cost
         ("turn90_4096", lambda: P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3), "al"))          # BASELINE configs[2]: its sweeps are over before anything splits
cases = cases + (
    # iteration caps that cut the streaks short (the remainder of a streak is min(inner, total) iterations, ilqr.hpp:600-611)
    ("obstacles_4096_caps", lambda: P.batch_three_obstacles(make, batch=4096, dtype=A.F32), "caps"),
    # MPC pattern behind a solve that split (al_solver.hpp:292-297): warm start, duals and penalties kept, the initial
    # state moved -- the shadow columns of the first solve are still in the arrays
    ("obstacles_4096_warm", lambda: P.batch_three_obstacles(make, batch=4096, dtype=A.F32), "warm"),
    # two handles in flight on one device (altro_solve_al_async), both splitting
    ("obstacles_4096_pair", lambda: P.batch_three_obstacles(make, batch=4096, dtype=A.F32), "pair"))
for name, fac, mode in cases:
    s = fac()
    ms = []
    if mode == "caps":
        s.set_options(max_iterations_inner=70, max_iterations_total=130)
    other = fac() if mode == "pair" else None
    for rep in range(3):   # (later solves reuse the shadow columns)
        s.reset_trajectory()
        if mode == "ilqr":
            s.rollout(); s.solve_ilqr()
        elif mode == "pair":
            other.reset_trajectory()
            s.solve_async(); other.solve_async(); s.wait(); other.wait()
            assert np.array_equal(s.get_trajectory()[0], other.get_trajectory()[0])
        else:
            s.solve()
        ms.append(s.get_timing()["total_ms"])
    if mode == "warm":
        X0 = s.get_trajectory()[0][:, 0, :].copy()
        s.set_options(reset_duals=0, initial_penalty=0.0)
        s.set_initial_state(X0 + 1e-3)
        s.solve()
    if other is not None:
        other.close()
    X, U = s.get_trajectory()
    st = s.get_stats()
    tm = s.get_timing()
    out[name + "_X"] = X; out[name + "_U"] = U
    K, d = s.get_gains()
    out[name + "_K"] = K; out[name + "_d"] = d
    out[name + "_lam"] = s.get_duals(); out[name + "_pen"] = s.get_penalties(); out[name + "_c"] = s.get_constraint_values()
    for f in st.dtype.names:
        out[name + "_st_" + f] = st[f]
    for k in (0, 50, 100):
        e = s.get_expansion(k)
        for key, v in e.items():
            if k < 100 or key in ("lxx", "lx"):   # (the terminal knot has no dynamics and no control blocks)
                out[name + "_exp%%d_%%s" %% (k, key)] = v
    out[name + "_costs"] = s.get_knot_costs()
    out[name + "_segcols"] = np.array([tm["segment_columns"]]); out[name + "_ms"] = np.array([min(ms[1:])])
    out[name + "_iters"] = np.array([tm["instance_iterations"], tm["sweep_launches"], tm["sweeps"]])
    s.close()
np.savez(sys.argv[1], **out)
'''


def test_segments_of_rejection_streaks_are_bit_identical(tmp_path):
    """Segments of rejection streaks in the batched sweeps (DevArrays::seg_*, forward_phase3): what is left of a
    100-iteration streak is split into four segments that run side by side in shadow columns; each column that reaches its
    segment's end compares its state, bit for bit, with what the next one assumed, and k_seg_fixup copies the last valid
    column back.  Every iteration is executed with the inputs the sequential order gives it, so NOTHING may differ from a
    solve without segments (ALTRO_HIP_SEGMENTS=0) -- while the chains of sweeps of BASELINE configs[3] get shorter."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # (third run: LDS, the candidate buffer and -- before every solve -- the shadow columns of every per-instance array full
    #  of NaN words: a clone that forgot to copy something, or a column read before it is rewritten, shows)
    for tag, env_extra in (("seg", {}), ("plain", {"ALTRO_HIP_SEGMENTS": "0"}), ("seg_poisoned", {"ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"})):
        out = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_SEG % root, out], check=True, env=dict(os.environ, **env_extra), timeout=900)
        res[tag] = np.load(out)
    a, b = res["seg"], res["plain"]
    for k in a.files:
        if not k.endswith(("_segcols", "_ms", "_iters")):
            assert np.array_equal(res["seg_poisoned"][k], b[k]), ("poisoned", k)
    assert res["seg_poisoned"]["obstacles_4096_r32_segcols"][0] > 1000
    for name in ("obstacles_4096_r32", "obstacles_2048_f64", "obstacles_1024_ilqr", "turn90_4096", "obstacles_4096_caps",
                 "obstacles_4096_warm", "obstacles_4096_pair"):
        print(name, "ms with / without segments", a[name + "_ms"][0], b[name + "_ms"][0], "shadow columns", a[name + "_segcols"][0],
              "(iterations, sweep launches, sweeps)", a[name + "_iters"], b[name + "_iters"])
        assert b[name + "_segcols"][0] == 0
    for k in a.files:
        if k.endswith(("_segcols", "_ms", "_iters")):
            continue
        assert np.array_equal(a[k], b[k]), (k, np.abs(np.asarray(a[k], float) - np.asarray(b[k], float)).max())
    # the streaks of configs[3] really were split, every iteration was executed, and the chains of sweeps got shorter
    assert a["obstacles_4096_r32_segcols"][0] > 1000 and a["obstacles_4096_caps_segcols"][0] > 500 and a["obstacles_4096_pair_segcols"][0] > 1000
    assert a["obstacles_4096_r32_iters"][0] == b["obstacles_4096_r32_iters"][0]
    assert a["obstacles_4096_r32_iters"][1] < 0.75 * b["obstacles_4096_r32_iters"][1]
    # (measured 32.7 - 33.3 against 35.8 - 36.2 ms; asserted loosely: a timing, the sweep count above is the structural check)
    assert a["obstacles_4096_r32_ms"][0] < 1.0 * b["obstacles_4096_r32_ms"][0], (a["obstacles_4096_r32_ms"], b["obstacles_4096_r32_ms"])


_SCRIPT_LOOP = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
cases = (("turn90_2304", lambda: P.batch_turn90(make, batch=2304, seed=P.SEED_BASE + 3)),       # more instances than slots: generations, refills
         ("turn90_640", lambda: P.batch_turn90(make, batch=640, seed=P.SEED_BASE + 5)),         # fits the slots
         ("obstacles_r32", lambda: P.batch_three_obstacles(make, batch=1100, dtype=A.F32)),   # fp32 records, circle constraints, global-source forward pass
         ("ilqr_800", lambda: P.batch_turn90(make, batch=800, seed=P.SEED_BASE + 7)))            # plain iLQR (no AL loop)
for name, fac in cases:
    s = fac()
    for rep in range(2):   # (the second solve reuses windows, control words and the tail list)
        s.reset_trajectory()
        if name.startswith("ilqr"):
            s.solve_ilqr()
        else:
            s.solve()
    X, U = s.get_trajectory()
    st = s.get_stats()
    tm = s.get_timing()
    K, d = s.get_gains()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_K"] = K; out[name + "_d"] = d
    out[name + "_lam"] = s.get_duals(); out[name + "_pen"] = s.get_penalties(); out[name + "_c"] = s.get_constraint_values()
    out[name + "_costs"] = s.get_knot_costs()
    for f in st.dtype.names:
        out[name + "_st_" + f] = st[f]
    for k in (0, 50, 100):
        e = s.get_expansion(k)
        for key, v in e.items():
            if k < 100 or key in ("lxx", "lx"):
                out[name + "_exp%%d_%%s" %% (k, key)] = v
    out[name + "_tm"] = np.array([tm["loop_workgroups"], tm["loop_instance_iterations"], tm["loop_handover"], tm["sweep_launches"]])
np.savez(sys.argv[1], **out)
'''


def test_device_side_sweep_loop_is_bit_identical(tmp_path):
    """k_sweep_loop (round 6): ONE launch of persistent workgroups runs the bulk phase -- expansions, backward pass, forward pass
    of the instances in their slots, iteration after iteration, slots refilled from a queue -- instead of the host-paced chains
    of sweeps, and hands what is left to the persistent tail kernel through a list the host never reads.  Same device code on
    the same inputs in the same per-instance order: trajectories, gains, multipliers, penalties, stored constraint values,
    expansion records, knot costs and every statistic must be those of the sweeps (ALTRO_HIP_SWEEP_LOOP=0), also with LDS and
    the shadow columns full of NaN words before every launch, and with one workgroup per CU instead of two."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, env_extra):
        out = str(tmp_path / f"loop_{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_LOOP % root, out], check=True, env=dict(os.environ, **env_extra), timeout=900)
        return np.load(out)
    ref = run("sweeps", {"ALTRO_HIP_SWEEP_LOOP": "0"})
    assert ref["turn90_2304_tm"][0] == 0 and ref["turn90_2304_tm"][3] > 8     # host-paced sweeps, no loop launch
    for tag, env in (("loop", {"ALTRO_HIP_SWEEP_LOOP": "1"}),
                     ("loop_poisoned", {"ALTRO_HIP_SWEEP_LOOP": "1", "ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"}),
                     ("loop_1_per_cu", {"ALTRO_HIP_SWEEP_LOOP": "1", "ALTRO_HIP_LOOP_PER_CU": "1"}),
                     ("loop_no_twins", {"ALTRO_HIP_SWEEP_LOOP": "1", "ALTRO_HIP_TWIN": "0"})):
        got = run(tag, env)
        for name in ("turn90_2304", "turn90_640", "obstacles_r32", "ilqr_800"):
            assert got[name + "_tm"][0] > 0 and got[name + "_tm"][3] == 1, (tag, name)   # the loop really ran, as one launch
            assert got[name + "_tm"][1] > 0
        for k in ref.files:
            if k.endswith("_tm"):
                continue
            assert np.array_equal(ref[k], got[k], equal_nan=True), (tag, k)


_SCRIPT_CTG_CHAINS = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
s = P.batch_turn90(make, batch=2304, seed=P.SEED_BASE + 3)
if sys.argv[2] == "record":
    s.set_record_ctg(True)
s.solve()
st = s.get_stats()
X, U = s.get_trajectory()
Pc, pc = s.get_ctg()
tm = s.get_timing()
np.savez(sys.argv[1], X=X, U=U, P=Pc, p=pc, it=st["iterations_total"], status=st["status"], launches=np.array([tm["sweep_launches"], tm["fused_sweeps"]]))
'''


def test_recording_the_cost_to_go_with_four_chains_of_sweeps(tmp_path):
    """altro_set_record_ctg(1) on a batch that runs as FOUR chains of sweeps (the first large handle of a process): the
    recording backward pass (k_backward_mfma<.., CTG>) of one chain runs beside the forward passes of the others, and must
    write nothing outside its own cost-to-go records.  Round 6 found its junk sink aliased on instance 0's line-search
    candidates (instance 0: 119 iterations instead of 11).  Statistics, trajectories and P, p must be those of the default
    solve (persistent kernel + replayed backward pass)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag in ("default", "record"):
        out = str(tmp_path / f"ctg_{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_CTG_CHAINS % root, out, tag], check=True, env=dict(os.environ, ALTRO_HIP_CHAINS="4"), timeout=600)
        res[tag] = np.load(out)
    a, b = res["default"], res["record"]
    assert b["launches"][1] == 0 and b["launches"][0] > 4 * 100    # four chains swept to the end, no persistent kernel
    for k in ("it", "status", "X", "U", "P", "p"):
        assert np.array_equal(a[k], b[k]), (k, np.flatnonzero((a[k] != b[k]).reshape(len(a[k]), -1).any(axis=1))[:8])


_SCRIPT_BIG_SRC = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, fac in (("quad12_r32", lambda: P.batch_quadrotor12(make, batch=300, dtype=A.F32)),
                  ("quad12_f64", lambda: P.batch_quadrotor12(make, batch=40, dtype=A.F64)),
                  ("tripleint", lambda: P.batch_triple_integrator(make, batch=200))):
    s = fac()
    if name == "tripleint":
        s.solve_ilqr()
    else:
        s.solve()
    X, U = s.get_trajectory(); st = s.get_stats(); K, d = s.get_gains()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_K"] = K; out[name + "_d"] = d
    out[name + "_lam"] = s.get_duals(); out[name + "_c"] = s.get_constraint_values()
    for f in st.dtype.names:
        out[name + "_st_" + f] = st[f]
np.savez(sys.argv[1], **out)
'''


def test_forward_pass_sources_of_the_large_models_are_bit_identical(tmp_path):
    """Round 6: the 12-state model's forward pass reads its rollout inputs from global memory (k_forward2<.., kSrcGlb>: one knot
    ahead, the gain record in its storage type, one barrier per knot, two workgroups per CU) instead of keeping K in global
    memory and the rest staged (kSrcKdg, rounds 2 - 5: one workgroup per CU).  Same arithmetic on the same values: the two
    variants must agree bit for bit, with fp32 and fp64 records; the triple integrator (n m = 12 as well) keeps its staged
    variant by default and must not change either way."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(tag, env_extra):
        out = str(tmp_path / f"bigsrc_{tag}.npz")
        subprocess.run([sys.executable, "-c", _SCRIPT_BIG_SRC % root, out], check=True, env=dict(os.environ, **env_extra), timeout=900)
        return np.load(out)
    ref = run("default", {})
    assert (ref["quad12_r32_st_status"] == 0).all()
    for tag, env in (("kdg", {"ALTRO_HIP_FWD_SRC": "kdg"}), ("global", {"ALTRO_HIP_FWD_SRC": "global"}),
                     ("global_poisoned", {"ALTRO_HIP_FWD_SRC": "global", "ALTRO_HIP_DEBUG_POISON": "7ff80000,mix"})):
        got = run(tag, env)
        for k in ref.files:
            assert np.array_equal(ref[k], got[k], equal_nan=True), (tag, k)
