"""A-trous launch time per step size for several BUILDS of the library, each in a process of its own (arms alternate, two rounds), and whether
every image the SVGF pass publishes is bit-identical to the first build's:   python scratch/ab_atrous_libs.py [lib.so ...]   (default: the in-tree library).
VHR_SIZE=WxH, VHR_SCENE=sponza_proc|bistro_proc, VHR_OPTS="k=v,k=v"."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path):
    import torch
    from vulkanhybridrenderer_amd import lib, scenes
    if path != "default":
        lib.LIB_PATH = os.path.abspath(path)
    from vulkanhybridrenderer_amd.harness import HybridFrameLoop
    W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
    scene = getattr(scenes, os.environ.get("VHR_SCENE", "sponza_proc"))()
    loop = HybridFrameLoop(scene, W, H, 12)
    ctx = loop.ctx
    ctx.set_option("svgf_async_unread", 0)
    for kv in filter(None, os.environ.get("VHR_OPTS", "").split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    for f in range(3):
        loop.frame(f)
    ctx.synchronize()
    pc = loop.path.push_constants()
    h = hashlib.sha256()
    for key in (lib.DENOISED, int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"]), int(pc["integrated_shadow_and_ao"][0]), int(pc["integrated_shadow_and_ao"][1])):
        h.update(ctx.download(key).tobytes())
    best = None
    for rep in range(3):
        ctx.set_kernel_timing(["svgf_atrous"]); ctx.kernel_time("svgf_atrous", reset=True)
        for f in range(3, 11):
            loop.frame(f)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time("svgf_atrous"); ctx.set_kernel_timing(False)
        us = ms / n * 1e3
        best = us if best is None else min(best, us)
    print(json.dumps({"lib": path, "atrous_us": round(best, 2), "hash": h.hexdigest()[:16], "fingerprint": lib.source_fingerprint()}), flush=True)
    loop.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    libs = sys.argv[1:] or ["default"]
    res = {l: [] for l in libs}
    hashes = {}
    for rnd in range(2):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(l, "FAILED", out.stderr[-1500:]); continue
            d = json.loads(line[-1]); res[l].append(d["atrous_us"]); hashes[l] = d["hash"]
    for l in libs:
        print(f"{l}: a-trous launch {min(res[l]) if res[l] else None} us {res[l]}  images {'== first' if hashes.get(l) == hashes.get(libs[0]) else 'DIFFER from first'} ({hashes.get(l)})", flush=True)
