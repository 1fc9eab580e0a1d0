#!/usr/bin/env python3
"""The driver's default sequence (bench.py --gpus N without --config): N = 1 is config 2, N > 1 is config 3's frame at 64 N spp in
cost-balanced bands.  Every rank's launch one after the other on ONE GPU, kernel times only (no gather): what the per-GPU rate of
an N-GPU run would be against the N = 1 line if nothing but the kernels counted.  usage (GPU box): python3 tools/weak_sequence.py"""
import importlib, os, statistics, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT)
srt=importlib.import_module("software-raytracer_amd"); stripes=importlib.import_module("software-raytracer_amd.stripes")
sc=srt.host.Scene(os.path.join(ROOT,"software-raytracer_amd","scenes","Scene1.json")); objs,n=sc.objects_copy()
W,H=1920,1080
def tracer():
    pt=srt.PathTracer(W,H); pt.set_scene(objs,n); pt.set_camera(srt.default_camera()); return pt
def ms(rows,spp):
    pt=tracer(); ts=[]
    for i in range(8):
        pt.render(spp=spp,bounces=8,seed=0,rows=rows); ts.append(pt.stats().kernel_ms)
    c=pt.stats().sample_chunks; pt.close(); return statistics.median(ts[3:]),c
pt=tracer(); rc=pt.estimate_row_costs(8,0); pt.close()
one,_=ms((0,H),32)
print("N=1: config 2 (32 spp) %.3f ms -> %.3e samples/s"%(one,W*H*32/one*1e3))
for N in (2,4,8):
    spp=64*N; bands=stripes.partition_rows(H,N,rc,align=2)
    t=[ms(b,spp) for b in bands]
    slow=max(x[0] for x in t); val=W*H*spp/slow*1e3
    print("N=%d spp %d: ms %s chunks %s | slowest %.2f mean/slowest %.3f | job %.3e samples/s, per GPU vs N=1: %.3f (kernel only, no gather)"%(N,spp," / ".join("%.2f"%x[0] for x in t),[x[1] for x in t],slow,sum(x[0] for x in t)/N/slow,val,val/N/(W*H*32/one*1e3)))
