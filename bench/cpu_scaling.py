"""Diagnose host-CPU scaling of the oracle (cpu_baseline leg): threads vs wall time."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    lg = int(sys.argv[2])
    mles = [orc.random_felts((1 << lg, 32), 42 + i) for i in range(8)]
    orc.CommittedRound([m[:256] for m in mles], 2)
    t0 = time.perf_counter(); orc.CommittedRound(mles, 2); dt = time.perf_counter() - t0
    print("threads=%s lg_rows=%d wall=%.3fs rows/s=%.0f" % (os.environ.get("OMP_NUM_THREADS"), lg, dt, (1 << lg) / dt))
else:
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f): print(f, open(f).read().strip())
    for t in (1, 8, 32, 64, 128, 256):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="false")
        subprocess.run([sys.executable, __file__, "child", "14" if t == 1 else "16"], env=env)
