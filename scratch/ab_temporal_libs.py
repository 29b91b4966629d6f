"""svgf.comp's launch on several builds of the library (each in its own process, arms alternated): python scratch/ab_temporal_libs.py lib.so ... ("default" = in-tree)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def child(path):
    import torch
    from vulkanhybridrenderer_amd import lib, scenes
    if path != "default": lib.LIB_PATH = os.path.abspath(path)
    from vulkanhybridrenderer_amd.harness import HybridFrameLoop
    W, H = [int(v) for v in os.environ.get("VHR_SIZE", "1920x1080").split("x")]
    loop = HybridFrameLoop(getattr(scenes, os.environ.get("VHR_SCENE", "sponza_proc"))(), W, H, 12)
    ctx = loop.ctx
    for f in range(3): loop.frame(f)
    ctx.synchronize()
    pc = loop.path.push_constants()
    h = hashlib.sha256()
    for key in (lib.DENOISED, int(pc["shadow_and_ao_history"]), int(pc["shadow_and_ao_moments_history"])): h.update(ctx.download(key).tobytes())
    best = 1e9
    for rep in range(3):
        ctx.set_kernel_timing(["svgf_temporal"]); ctx.kernel_time("svgf_temporal", reset=True)
        for f in range(3, 11): loop.frame(f)
        torch.cuda.synchronize()
        ms, n = ctx.kernel_time("svgf_temporal"); ctx.set_kernel_timing(False)
        best = min(best, ms / n * 1e3)
    print(json.dumps({"lib": path, "us": round(best, 2), "hash": h.hexdigest()[:16]}), flush=True)
    loop.close()
if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    libs = sys.argv[1:] or ["default"]
    res, hashes = {l: [] for l in libs}, {}
    for rnd in range(2):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            if not line: print(l, "FAILED", out.stderr[-800:]); continue
            d = json.loads(line[-1]); res[l].append(d["us"]); hashes[l] = d["hash"]
    for l in libs:
        print(f"{l}: svgf.comp launch {min(res[l]) if res[l] else None} us {res[l]}  images {'== first' if hashes.get(l) == hashes.get(libs[0]) else 'DIFFER'}", flush=True)
