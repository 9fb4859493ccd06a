"""Thread scaling of the CPU oracle on this host (bench.py's cpu_baseline leg): iterations/s, wall vs CPU time of the
team for 1 .. all hardware threads, and the container's CPU quota.   python scripts/probe_cpu_scaling.py [batch]"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
A = bench.graft.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
lib = bench.load_oracle()
lib.oracle_bench_cpu_seconds.restype = ctypes.c_double
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat", "/proc/loadavg"):
    try:
        print(f, "=", open(f).read().strip().replace("\n", " | "))
    except Exception as e:
        print(f, "unreadable:", type(e).__name__)
hw, phys = lib.oracle_host_threads(), lib.oracle_host_physical_cores()
print("hardware threads", hw, "physical cores", phys)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
omake = lambda n_, m_, N_, b_, d_: A.BatchSolver(n_, m_, N_, b_, d_, _lib=lib, _prefix="oracle_")
o = P.batch_turn90(omake, batch=B, dtype=A.F64, seed=P.SEED_BASE + 3)
lib.oracle_prepare(o._h)
lib.oracle_set_threads(o._h, hw)
lib.oracle_bench_al(o._h, 1)
it = float(o.get_stats()["iterations_total"].sum())
nts = sorted({1, 2, 4, 8, 16, 32, 64, phys, hw} & set(range(1, hw + 1)))
base = None
for nt in nts:
    lib.oracle_set_threads(o._h, nt)
    stride = max(1, B // (64 * nt))  # ~64 solves per thread: keeps the small teams short
    lib.oracle_set_ilqr_mode(o._h, 0)
    lib.oracle_bench_subset(o._h, stride, 1)
    sec = lib.oracle_bench_seconds(o._h)
    cpu = lib.oracle_bench_cpu_seconds(o._h)
    st = o.get_stats()["iterations_total"][::stride].sum()
    rate = st / sec
    base = base or rate
    print(f"threads {nt:4d}: {rate:10.0f} iterations/s  x{rate / base:6.1f}  wall {sec:.3f} s  team CPU time {cpu:.3f} s "
          f"= {cpu / (sec * nt):.2f} of threads x wall  busy min/max {lib.oracle_bench_busy(o._h, 0):.3f}/{lib.oracle_bench_busy(o._h, 1):.3f}")
