import sys, os, torch
sys.path.insert(0, os.getcwd())
from cdlrm_amd import ops, synth
DEV=torch.device("cuda:0")
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
ln=synth.TERABYTE_COUNTS; T=len(ln); D=128
for B in (512,1024,2048,4096,8192,16384,65536):
    cs=[min(n,150001) for n in ln]
    ctx=ops.CacheCtx(ln,cs,D,16,B,DEV)
    slots=torch.stack([torch.randint(0, 16*c, (B,), device=DEV, dtype=torch.int32) for c in cs])
    work=ops.embbag_bwd_work(ctx,B,DEV)
    us=timeit(lambda: ops.embbag_bwd_prepare(ctx,slots,work))
    print("n=%6d  prepare %.1f us" % (B, us))
