"""Streams a 512 MiB RGBA16F image once with 4-, 8- and 16-byte lanes (tools/profile_traffic.sh runs this under
rocprofv3 --pmc FETCH_SIZE): known bytes / reported bytes = the FETCH_SIZE correction for each access width."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkanhybridrenderer_amd import abi, lib   # noqa: E402

W, H = 8192, 8192                    # 8 B/px -> 512 MiB: larger than the 256 MiB Infinity Cache
ctx = lib.Context(W, H)
img = ctx.upload_new_storage_image(W, H, abi.FORMAT_R16G16B16A16_SFLOAT)
ctx.synchronize()
for width in (4, 8, 16, 4, 8, 16):
    ctx.calibration_stream_read(img, width)
    ctx.synchronize()
print("calibration bytes per launch", W * H * 8)
ctx.close()
