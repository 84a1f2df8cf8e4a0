"""cv-mode and own-estimator PnP call times (600 correspondences, 30 % outliers).  usage: python tools/time_pnp.py"""
import sys,time,numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,"dynamic-visual-slam_amd")); sys.path.insert(0,os.path.join(ROOT,"tests"))
import ransac_scenes as rs
from dvslam_amd import FrontendGlue
g=FrontendGlue()
sc=rs.two_view(600,0.3,0.5,1)
for name,fn in (("cv",lambda: g.solve_pnp_ransac_cv(sc["X"],sc["pts2"],sc["K4"])),("own",lambda: g.solve_pnp_ransac(sc["X"],sc["pts2"],sc["K4"]))):
    fn(); t0=time.perf_counter()
    for _ in range(50): fn()
    print(name,"us per call",(time.perf_counter()-t0)/50*1e6)
