"""print psxhip_mdec_query_geometry for the sizes the docs quote"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psxavenc_amd import _lib


class Geo(C.Structure):
    _fields_ = [("fits", C.c_int32), ("groups_per_cu", C.c_int32), ("wavefronts_per_group", C.c_int32), ("frames_in_flight", C.c_int32),
                ("max_frame_size_limit", C.c_int32), ("image_tile_bytes", C.c_int32), ("lds_bytes_per_group", C.c_int64), ("lds_bytes_per_cu", C.c_int64)]


for (w, h, b) in ((320, 240, 8192), (320, 240, 18144), (640, 480, 8192), (640, 480, 32768), (640, 512, 20160), (160, 112, 4096), (16, 16, 64)):
    g = Geo()
    _lib.check(_lib.lib().psxhip_mdec_query_geometry(0, 0, w, h, b, C.byref(g)))
    print("%dx%d budget %d: fits %d, %d group(s)/CU x %d wavefronts, %d B LDS/group, frames in flight %d, largest budget %d" % (
        w, h, b, g.fits, g.groups_per_cu, g.wavefronts_per_group, g.lds_bytes_per_group, g.frames_in_flight, g.max_frame_size_limit))
