"""Host-only: which frame "bvh_frame" 1 finds for a scene, and what it does to the tree (no GPU)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from vulkanhybridrenderer_amd import lib, scenes
for name in sys.argv[1:]:
    sc = getattr(scenes, name)()
    for mode in (0, 1):
        c = lib.Context(64, 64, host_only=True)
        c.set_option("bvh_frame", mode)
        t = time.time(); c.update_geometry(sc.vertices, sc.indices, sc.primitives); dt = time.time() - t
        print(name, mode, c.bvh_statistics(), f"{dt:.2f}s", np.round(c.bvh_frame(), 4).tolist(), flush=True)
        c.close()
    if sc.camera.get("world") is not None:
        print("   the scene's own rotation, transposed:", np.round(np.asarray(sc.camera["world"])[:3, :3].T, 4).tolist())
