#!/usr/bin/env python3
"""What leaves the map at cfg3 size (10 000^2 cells, 50 M points): ms per layer -> image (kernel /
incl. download), GeoTiff write, grid_map_msgs message straight from the device, binary cloud load.
    python tools/bench_export.py [side]"""
import os, sys, time, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import aerial_mapper_amd as A
from aerial_mapper_amd import export as E, hip_lib as L, synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
res = 0.25
out = {"side": side}
lib = L.load()
dev = torch.device("cuda", 0)
with A.HostSession(A.GridMapSettings(0.0, 0.0, side * res, side * res, res)) as hs:
    n = side * side // 2
    pts = synth.make_points_torch(n, (side * res / 2 + 3, side * res / 2 + 3), 43, dev).cpu().numpy()
    hs.dsm_process(A.DsmSettings(1), pts)
    # layer -> image, device only (kernel) and to the host
    ctx = lib.amhip_session_context(hs._h, 0)
    img_dev = torch.empty((side, side), dtype=torch.uint8, device=dev)
    for _ in range(2):
        L.check(lib.amhip_layer_to_image_dev(ctx, L.LAYER_ELEVATION, 0, 390.0, 410.0,
                                             C.c_void_p(img_dev.data_ptr()), side))
    L.check(lib.amhip_ctx_synchronize(ctx))
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        L.check(lib.amhip_layer_to_image_dev(ctx, L.LAYER_ELEVATION, 0, 390.0, 410.0,
                                             C.c_void_p(img_dev.data_ptr()), side))
    L.check(lib.amhip_ctx_synchronize(ctx))
    out["layer_to_image_kernel_ms"] = round((time.perf_counter() - t0) / K * 1e3, 3)
    img = E.session_layer_to_image(hs, "elevation", 390.0, 410.0)
    t0 = time.perf_counter()
    img = E.session_layer_to_image(hs, "elevation", 390.0, 410.0)
    out["layer_to_image_to_host_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    d = tempfile.mkdtemp()
    t0 = time.perf_counter()
    E.write_geotiff(os.path.join(d, "dsm.tif"), img, (0.0, res, 0.0, 0.0, 0.0, -res))
    out["geotiff_write_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    msg = E.session_grid_map_msg(hs, 1)
    t0 = time.perf_counter()
    msg = E.session_grid_map_msg(hs, 2, out=msg)      # (a publisher reuses its buffer)
    out["grid_map_msg_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["grid_map_msg_bytes"] = int(msg.nbytes)
    f = os.path.join(d, "cloud.ampc")
    E.write_point_cloud_binary(f, pts)
    c = E.load_point_cloud_binary(f); c.close()
    t0 = time.perf_counter()
    c = E.load_point_cloud_binary(f)
    out["binary_cloud_load_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["binary_cloud_points"] = c.n
    out["binary_cloud_GBs"] = round(24.0 * c.n / (time.perf_counter() - t0) / 1e9, 1)
    c.close()
print(json.dumps(out))
