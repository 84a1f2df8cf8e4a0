import sys, numpy as np
sys.path.insert(0,'dynamic-visual-slam_amd'); sys.path.insert(0,'tests')
import torch
from dvslam_amd import synth, BAProblem
import oracle_bindings as ob, ba_bracket as bb
np.set_printoptions(linewidth=250, precision=10)
for kw in bb.HARD:
    P=synth.make_ba_problem(**kw)
    a=BAProblem(P); b=BAProblem(P); o=ob.OracleBA(P)
    sa=a.solve(40); sb=b.solve_device(40); so=o.solve(40)
    ta,tb,to=a.trace(),b.trace(),o.trace()
    print(kw['K'],kw['L'],kw['seed'],'host',(sa.termination,sa.num_successful_steps,sa.num_iterations,sa.final_cost),'dev',(sb.termination,sb.num_successful_steps,sb.num_iterations,sb.final_cost),'orc',(so.termination,so.num_successful_steps,so.num_iterations,so.final_cost))
    n=min(len(ta),len(tb),len(to))
    for i in range(n):
        print(i, int(ta[i,1]),int(tb[i,1]),int(to[i,1]), '%.6e %.6e %.6e'%(ta[i,0],tb[i,0],to[i,0]), '%.9e %.9e %.9e'%(ta[i,5],tb[i,5],to[i,5]), 'rho %.4f %.4f %.4f'%(ta[i,4],tb[i,4],to[i,4]))
