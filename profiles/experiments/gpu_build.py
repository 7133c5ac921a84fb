"""Scene hand-over time (mirt_set_scene) and trace throughput: host SAH sweep vs GPU LBVH (policy.gpu_build) for the internal tree."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mirt = importlib.import_module("cpu-raytracing-experiments_amd")
for n, w, h, mb in ((1000, 1024, 1024, 5), (10000, 1920, 1088, 9), (100000, 2048, 1024, 9)):
    sc = mirt.scene.synthetic(n, ambient=0.5 if n == 1000 else 0.0)
    for gpu_build in (False, True):
        r = mirt.Renderer(sc, max_bounces=mb, use_bvh=True, gpu_build=gpu_build, count_traffic=True)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r.UpdateScene(); ts.append(time.perf_counter() - t0)     # reference-order builder (host) + mirt_set_scene
        tb = []
        for _ in range(3):
            t0 = time.perf_counter(); mirt.bvh_build(r.geometry); mirt.light_list(r.geometry, r.material); tb.append(time.perf_counter() - t0)   # the caller's (reference-order) tree alone
        r.Resize(w, h); r.Accumulate(8); c = r.counters()
        r.set_policy(count_traffic=0); r.Accumulate(16)
        t0 = time.perf_counter(); r.Accumulate(32); dt = time.perf_counter() - t0
        rays = r.counters()["rays"]
        print(f"S({n}) gpu_build={int(gpu_build)}: UpdateScene {min(ts)*1e3:.1f} ms (reference-order host builder {min(tb)*1e3:.1f} ms, mirt_set_scene {max(min(ts)-min(tb),0)*1e3:.1f} ms) | boxes/ray {c['nodes']/c['rays']:.1f} shadow {c['shadow_nodes']/max(c['shadow_rays'],1):.1f} | depth {r.debug_info()['depth']} | "
              f"{32*w*h/dt/1e6:.0f} Mpath/s", flush=True)
        r.close()
