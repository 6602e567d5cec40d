import time, numpy as np, os, math, sys
sys.path.insert(0,'.')
os.environ.setdefault("OMP_PROC_BIND","close"); os.environ.setdefault("OMP_PLACES","cores")
import oracle, bench
o=oracle.get("f32"); o.set_parallel(True)
n,L=1000000,107.7217345
pos=bench.lattice(n,L,1234); vel=np.zeros((n,3),np.float32); force=np.zeros((n,4),np.float32)
par=o.lj_params(2.5,1.0,1.0)
def T(f,*a,**k):
    t=time.perf_counter(); r=f(*a,**k); return r,time.perf_counter()-t
cd,oL,oper=o.celllist_create_grid(L,1,2.5)
for rep in range(3):
    cl,t1=T(o.celllist_build,pos,oL,oper,cd)
    (f,_,_),t2=T(o.lj_transverse_celllist,cl,L,1,par,1,n)
    _,t3=T(o.verletnvt_gj,1,pos,vel,f,0.005,1.0,0.1,1,1234)
    _,t4=T(o.verletnvt_gj,2,pos,vel,f,0.005,1.0,0.1,1,1234)
    print("build %.3f traverse %.3f gj1 %.3f gj2 %.3f"%(t1,t2,t3,t4), "cores",os.cpu_count(), len(os.sched_getaffinity(0)))
for th in (32,64,128):
    os.environ["OMP_NUM_THREADS"]=str(th)
