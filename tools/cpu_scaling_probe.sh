mkdir -p gpurun_out; echo "nproc=$(nproc) affinity=$(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)" 
python3 - <<'PY'
import ctypes, numpy as np, sys, time, os
sys.path.insert(0,'.')
import libdvd_audio_amd as pkg
from tests import oracle_lib
syn=pkg.synth
cfg=syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=512)
flat,offs,sizes,frames=syn.batch(cfg,1,512)
for kind,lib in (("reference",oracle_lib.Reference().lib),("port",oracle_lib.Oracle().lib)):
    fn=lib.cpu_pool_decode
    fn.restype=ctypes.c_ulong
    fn.argtypes=[ctypes.c_void_p]*3+[ctypes.c_uint32]+[ctypes.c_uint]*4+[ctypes.c_void_p,ctypes.c_size_t,ctypes.c_uint,ctypes.c_double,ctypes.POINTER(ctypes.c_double)]
    pcm=np.empty((512,6,40960),np.int32)
    o=np.ascontiguousarray(offs,np.uint64); s=np.ascontiguousarray(sizes,np.uint64)
    secs=ctypes.c_double()
    for th in (1,2,4,8,16,32,64,128,256):
        n=min(512,max(8,th*4))
        d=fn(flat.ctypes.data,o.ctypes.data,s.ctypes.data,n,2,1,12,6,pcm.ctypes.data,40960,th,2.0,ctypes.byref(secs))
        print(kind,"threads",th,"Msamples/s",round(d*40960*6/secs.value/1e6,1), flush=True)
PY
