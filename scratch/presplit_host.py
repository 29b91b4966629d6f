"""Host-only: what bvh_presplit does to the reference count / tree of a scene (no GPU)."""
import sys, time
sys.path.insert(0, ".")
from vulkanhybridrenderer_amd import lib, scenes
for name in sys.argv[1:]:
    name, _, pct = name.partition(":")
    sc = getattr(scenes, name)()
    c = lib.Context(64, 64, host_only=True)
    c.set_option("bvh_presplit", int(pct or 0)); c.set_option("bvh_frame", 0)
    t = time.time(); c.update_geometry(sc.vertices, sc.indices, sc.primitives); dt = time.time() - t
    print(name, pct, c.bvh_statistics(), f"{dt:.2f}s", flush=True)
    c.close()
