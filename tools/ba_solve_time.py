import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dvslam_amd, oracle_bindings as ob
from dvslam_amd import synth
for K, L, it in [(5, 500, 20), (10, 2000, 20)]:
    P = synth.make_ba_problem(K=K, L=L, seed=42)
    g = dvslam_amd.BAProblem(P); g.evaluate()
    t0 = time.perf_counter(); s = g.solve(it); dt = time.perf_counter() - t0
    o = ob.OracleBA(P); t0 = time.perf_counter(); s2 = o.solve(it); dt2 = time.perf_counter() - t0
    print(f"K={K} L={L}: GPU-evaluated LM {1e3*dt:.1f} ms ({s.num_iterations} it, final {s.final_cost:.3f}) | oracle CPU LM {1e3*dt2:.1f} ms ({s2.num_iterations} it, final {s2.final_cost:.3f})")
