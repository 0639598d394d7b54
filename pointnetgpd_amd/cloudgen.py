"""Drop-in (same names, arguments and return layouts) for the two hot functions of the reference's 36-hour
preprocessing script ``PointNetGPD/ycb_cloud_generate.py`` (README.md:166): ``registerDepthMap`` (:60-121) and
``registeredDepthMapToPointCloud`` (:124-184) — pure-Python double loops over 480x640 pixels there, one kernel launch
each here (``pointnetgpd_amd/csrc/pngpd_gpd.hip``).  numpy in, numpy out, bit-identical values.

One deliberate difference: a NaN depth is SKIPPED by ``registeredDepthMapToPointCloud`` here (the kernel keeps pixels
with ``depth > 0``), whereas the reference's test ``depth <= 0`` is False for NaN and it would emit a NaN point.  The
YCB depth maps hold zeros for missing returns, never NaN."""
import numpy as np

from . import gpd_ops


def registerDepthMap(unregisteredDepthMap, rgbImage, depthK, rgbK, H_RGBFromDepth):
    return gpd_ops.register_depth_map(np.asarray(unregisteredDepthMap, dtype=np.float64), rgbImage.shape, depthK, rgbK,
                                      H_RGBFromDepth).cpu().numpy()


def registeredDepthMapToPointCloud(depthMap, rgbImage, rgbK, refFromRGB, objFromref, organized=False):
    if organized:      # (h,w,6): NaN xyz / zero colour where depth <= 0 (:147-155); never used by the reference's generate() (:370)
        return gpd_ops.depth_map_to_cloud_organized(np.asarray(depthMap, dtype=np.float64), rgbK, refFromRGB, objFromref,
                                                    np.asarray(rgbImage)).cpu().numpy()
    xyz, col = gpd_ops.depth_map_to_cloud(np.asarray(depthMap, dtype=np.float64), rgbK, refFromRGB, objFromref,
                                          rgb=np.asarray(rgbImage))
    cloud = np.empty((1, xyz.shape[0], 6))
    cloud[0, :, :3] = xyz.cpu().numpy()
    cloud[0, :, 3:] = col.cpu().numpy()
    return cloud
