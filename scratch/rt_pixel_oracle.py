import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from vulkanhybridrenderer_amd import scenes, camera
sc = scenes.rotated(scenes.sponza_proc(0.3), rot_y=0.6, rot_x=0.25)
Wv, Hv = 300, 170
pfds = camera.dolly_frames(sc, Wv, Hv, 2)
ob.build()
osc = ob.Scene(sc)
img, rays = osc.raytraced(pfds[1], Wv, Hv, False)
print("oracle pixel (165,118):", img[118, 165].tolist(), "neighbours", img[118, 163:168].tolist(), "rays", rays)
