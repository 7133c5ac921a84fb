"""mirt — MI355X-native wavefront path tracer for the reference's Renderer::Accumulate hot path.

This package is the thin Python host layer over the C-ABI in ``include/mirt.h`` (implemented by the
hand-written HIP kernels in ``csrc/``).  ``Renderer`` mirrors the reference's ``Renderer<Policy>``
interface (Renderer.hpp:51-68,73,436): ``Resize``, ``ResetAccumulator``, ``Accumulate``, ``Render``,
``GetFrame`` with the same meaning; the scene is handed over in the reference's byte layouts.

There is no CPU fallback: if ``libmirt.so`` or a gfx950 device is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import distributed as distributed  # noqa: F401
from . import hdr as hdr  # noqa: F401
from . import scene as scene  # noqa: F401  (re-export)
from .scene import MATERIAL, NODE, SPHERE, Camera, Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmirt.so")

MIRT_OK, MIRT_NOT_READY = 0, 1
KERNEL_CLASSES = ("raygen", "trace", "shade", "shadow", "resolve")


class MirtError(RuntimeError):
    pass


class Policy(C.Structure):
    """mirt_policy — RendererPolicy (Renderer.hpp:19-26) + the path's compile-time switches."""
    _fields_ = [("max_bounces", C.c_uint32), ("buckets", C.c_uint32), ("mis", C.c_uint32), ("use_bvh", C.c_uint32),
                ("count_traffic", C.c_uint32), ("profile", C.c_uint32), ("max_batch", C.c_uint32), ("reference_tree", C.c_uint32),
                ("streams", C.c_uint32), ("gpu_build", C.c_uint32), ("trace_primary_rays", C.c_uint32), ("_reserved", C.c_uint32 * 1)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "shadow_rays", "nodes", "spheres", "shadow_nodes", "shadow_spheres", "terminated", "dropped")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 5), ("launches", C.c_uint64 * 5)]


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libmirt.so and declare every symbol of include/mirt.h.  Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirtError(f"{LIB_PATH} not built — run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    P, u32, i32, f, vp = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_void_p
    sigs = {
        "mirt_create": [i32, C.POINTER(P)],
        "mirt_destroy": [P],
        "mirt_bvh_build": [vp, u32, vp, C.POINTER(u32), vp],
        "mirt_light_list": [vp, u32, vp, u32, vp, C.POINTER(u32)],
        "mirt_set_scene": [P, vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, vp, u32, u32],
        "mirt_set_camera": [P, vp, vp, f, f, f, f],
        "mirt_set_policy": [P, C.POINTER(Policy)],
        "mirt_get_policy": [P, C.POINTER(Policy)],
        "mirt_resize": [P, u32, u32],
        "mirt_set_tile_range": [P, u32, u32], "mirt_set_tile_rows": [P, u32, u32],
        "mirt_reset": [P],
        "mirt_accumulate": [P, u32],
        "mirt_accumulate_async": [P, u32],
        "mirt_synchronize": [P],
        "mirt_get_accumulations": [P, C.POINTER(u32)],
        "mirt_accumulator_floats": [P, C.POINTER(C.c_size_t)],
        "mirt_read_accumulator": [P, vp],
        "mirt_accumulator_device": [P, C.POINTER(vp), C.POINTER(C.c_size_t)],
        "mirt_load_accumulator": [P, vp, i32, u32],
        "mirt_render": [P, vp],
        "mirt_get_counters": [P, C.POINTER(Counters)],
        "mirt_get_kernel_times": [P, C.POINTER(KernelTimes), i32],
        "mirt_get_stream": [P, C.POINTER(vp)],
        "mirt_debug_raygen": [P, u32, vp, vp],
        "mirt_debug_trace_closest": [P, C.c_size_t, vp, vp, vp, vp],
        "mirt_debug_trace_shadow": [P, C.c_size_t, vp, vp, vp, vp],
        "mirt_debug_math": [P, i32, C.c_size_t, vp, vp],
        "mirt_debug_info": [P, vp],
        "mirt_debug_primary_lists": [P, vp],
        "mirt_debug_allow_half_boxes": [P, i32],
    }
    G = C.c_void_p
    sigs.update({
        "mirt_group_create": [vp, i32, C.POINTER(G)],
        "mirt_group_destroy": [G],
        "mirt_group_size": [G, C.POINTER(i32)],
        "mirt_group_member": [G, i32, C.POINTER(P)],
        "mirt_group_set_scene": [G, vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, vp, u32, u32],
        "mirt_group_set_camera": [G, vp, vp, f, f, f, f],
        "mirt_group_set_policy": [G, C.POINTER(Policy)],
        "mirt_group_resize": [G, u32, u32],
        "mirt_group_reset": [G],
        "mirt_group_accumulate": [G, u32],
        "mirt_group_accumulate_async": [G, u32],
        "mirt_group_synchronize": [G],
        "mirt_group_get_accumulations": [G, C.POINTER(u32)],
        "mirt_group_get_counters": [G, C.POINTER(Counters)],
        "mirt_group_gather": [G],
        "mirt_group_last_gather_ms": [G, C.POINTER(C.c_double)],
        "mirt_group_accumulator_floats": [G, C.POINTER(C.c_size_t)],
        "mirt_group_read_accumulator": [G, vp],
        "mirt_group_render": [G, vp],
        "mirt_group_rccl_selftest": [i32, C.c_size_t],
    })
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mirt_last_error.argtypes = [P]
    lib.mirt_last_error.restype = C.c_char_p
    lib.mirt_group_last_error.argtypes = [P]
    lib.mirt_group_last_error.restype = C.c_char_p
    lib._declared = tuple(sigs) + ("mirt_last_error", "mirt_group_last_error")
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def bvh_build(geometry: np.ndarray):
    """BoundingVolumeHierarchy<Sphere> ctor (BVH.hpp:90-206) -> (nodes, bvh-order prims)."""
    lib = load_library()
    geometry = np.ascontiguousarray(geometry, dtype=SPHERE)
    n = len(geometry)
    nodes = np.zeros(max(2 * n, 1), dtype=NODE)
    prims = np.zeros(n, dtype=SPHERE)
    n_nodes = C.c_uint32(0)
    rc = lib.mirt_bvh_build(_ptr(geometry), n, _ptr(nodes), C.byref(n_nodes), _ptr(prims))
    if rc != MIRT_OK:
        raise MirtError(f"mirt_bvh_build failed ({rc})")
    return nodes[: n_nodes.value].copy(), prims


def light_list(geometry: np.ndarray, material: np.ndarray) -> np.ndarray:
    """LightingAcceleration ctor (Scene.hpp:12-16)."""
    lib = load_library()
    geometry = np.ascontiguousarray(geometry, dtype=SPHERE)
    material = np.ascontiguousarray(material, dtype=MATERIAL)
    out = np.zeros(max(len(geometry), 1), dtype=np.int32)
    n = C.c_uint32(0)
    rc = lib.mirt_light_list(_ptr(geometry), len(geometry), _ptr(material), len(material), _ptr(out), C.byref(n))
    if rc != MIRT_OK:
        raise MirtError(f"mirt_light_list failed ({rc})")
    return out[: n.value].copy()


class Renderer:
    """Mirror of the reference's ``Renderer<Policy>`` (Renderer.hpp:28-68) over the C-ABI.

    ``Renderer(scene)`` keeps a reference to the scene like the original; call ``UpdateScene`` after
    editing it (the reference rebuilds BVH + light list and resets, Application.cpp:508-510).
    """

    def __init__(self, scene: Scene, device: int = 0, max_bounces: int = 16, buckets: int = 5, mis: bool = True,
                 use_bvh: bool = False, count_traffic: bool = False, profile: bool = False, max_batch: int = 0,
                 allow_half_boxes: bool = True, reference_tree: bool = False, streams: int = 0, gpu_build: bool = False, trace_primary_rays: bool = False):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.mirt_create(device, C.byref(self._ctx))
        if rc != MIRT_OK:
            raise MirtError(f"mirt_create failed ({rc}): {self._lib.mirt_last_error(None).decode()}")
        self.scene = scene
        self.width = self.height = 0
        self.framebuffer = None
        self.policy = Policy(max_bounces, buckets, int(mis), int(use_bvh), int(count_traffic), int(profile), max_batch, int(reference_tree), int(streams), int(gpu_build), int(trace_primary_rays))
        self._check(self._lib.mirt_set_policy(self._ctx, C.byref(self.policy)))
        self._check(self._lib.mirt_debug_allow_half_boxes(self._ctx, int(allow_half_boxes)))
        self.UpdateScene()

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise MirtError(f"mirt call failed ({rc}): {self._lib.mirt_last_error(self._ctx).decode()}")
        return rc

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.mirt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_policy(self) -> dict:
        """The policy in effect (max_batch / streams resolved where they were left at 0 = auto)."""
        p = Policy()
        self._check(self._lib.mirt_get_policy(self._ctx, C.byref(p)))
        return {name: int(getattr(p, name)) for name, _ in Policy._fields_ if not name.startswith("_")}

    def set_policy(self, **kw):
        p = Policy.from_buffer_copy(self.policy)
        for k, v in kw.items():
            setattr(p, k, int(v))
        self._check(self._lib.mirt_set_policy(self._ctx, C.byref(p)))
        self.policy = p                                    # only a policy the library accepted becomes this object's

    # -- scene hand-over (Application.cpp:230-234) -------------------------------------------------
    def UpdateScene(self, nodes=None):
        """Hands the scene over again (after an edit).  `nodes`: a caller-made tree over the reference-order prims instead of the
        reference builder's (BVH.hpp:18-31 layout, children at first_id / first_id+1, leaves first_id..first_id+prim_count-1)."""
        s = self.scene
        self.geometry = np.ascontiguousarray(s.geometry, dtype=SPHERE)
        self.material = np.ascontiguousarray(s.material, dtype=MATERIAL)
        self.nodes, self.prims = bvh_build(self.geometry)
        if nodes is not None:
            self.nodes = np.ascontiguousarray(nodes, dtype=NODE)
        self.lights = light_list(self.geometry, self.material)
        hdri = np.ascontiguousarray(s.hdri, dtype=np.float32)
        amb = np.ascontiguousarray(s.ambient, dtype=np.float32)
        lights = self.lights if len(self.lights) else np.zeros(1, dtype=np.int32)
        self._check(self._lib.mirt_set_scene(self._ctx, _ptr(self.geometry), _ptr(self.prims), len(self.geometry), _ptr(self.nodes), len(self.nodes),
                                             _ptr(self.material), len(self.material), _ptr(lights), len(self.lights), _ptr(amb),
                                             _ptr(hdri), hdri.shape[1], hdri.shape[0]))
        self.UpdateCamera()

    def UpdateCamera(self):
        cam: Camera = self.scene.camera
        pos = np.ascontiguousarray(cam.pos, dtype=np.float32)
        ori = np.ascontiguousarray(cam.orient, dtype=np.float32)
        self._check(self._lib.mirt_set_camera(self._ctx, _ptr(pos), _ptr(ori), float(cam.half_width), float(cam.half_height), float(cam.z), float(cam.exposure)))

    # -- the reference interface ---------------------------------------------------------------
    @staticmethod
    def RequiredTiling() -> int:          # Renderer.hpp:36
        return 16

    def Resize(self, new_width: int, new_height: int):     # Renderer.hpp:53-63 (+ camera.Resize, Application.cpp:375-376)
        self.width, self.height = int(new_width), int(new_height)
        self.scene.camera.resize(self.width, self.height)
        self.UpdateCamera()
        self._check(self._lib.mirt_resize(self._ctx, self.width, self.height))
        self.framebuffer = np.zeros((self.height, self.width, 4), dtype=np.float32)

    def ResetAccumulator(self):            # Renderer.hpp:64-67
        self._check(self._lib.mirt_reset(self._ctx))

    def Accumulate(self, n_calls: int = 1):   # Renderer.hpp:73-434
        self._check(self._lib.mirt_accumulate(self._ctx, n_calls))

    def AccumulateAsync(self, n_calls: int = 1):
        self._check(self._lib.mirt_accumulate_async(self._ctx, n_calls))

    def Synchronize(self):
        self._check(self._lib.mirt_synchronize(self._ctx))

    def Render(self) -> bool:              # Renderer.hpp:436-478; False = not a multiple of `buckets` yet, frame unchanged
        rc = self._check(self._lib.mirt_render(self._ctx, _ptr(self.framebuffer)))
        return rc == MIRT_OK

    def GetFrame(self) -> np.ndarray:      # Renderer.hpp:68 — RGBA32F rows, row 0 = y 0 (bottom on screen)
        return self.framebuffer

    # -- sharding / state access -------------------------------------------------------------------
    def SetTileRange(self, first_tile: int, n_tiles: int):
        self._check(self._lib.mirt_set_tile_range(self._ctx, first_tile, n_tiles))

    def SetTileRows(self, first_row: int, row_stride: int):
        """Multi-GPU sharding by interleaved tile rows: this context renders tile rows first_row, first_row + row_stride, ..."""
        self._check(self._lib.mirt_set_tile_rows(self._ctx, first_row, row_stride))

    @property
    def accumulations(self) -> int:
        v = C.c_uint32(0)
        self._check(self._lib.mirt_get_accumulations(self._ctx, C.byref(v)))
        return v.value

    def accumulator(self) -> np.ndarray:
        """[local tile][bucket][rgb][256] f32 — AccumulationTile layout (Renderer.hpp:43-46)."""
        n = C.c_size_t(0)
        self._check(self._lib.mirt_accumulator_floats(self._ctx, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self._lib.mirt_read_accumulator(self._ctx, _ptr(out)))
        return out.reshape(n.value // (self.policy.buckets * 768), self.policy.buckets, 3, 256)      # (a context may own no tile at all)

    def accumulator_device(self):
        p, b = C.c_void_p(), C.c_size_t(0)
        self._check(self._lib.mirt_accumulator_device(self._ctx, C.byref(p), C.byref(b)))
        return p.value, b.value

    def load_accumulator(self, src, accumulations: int, is_device: bool = False):
        ptr = C.c_void_p(src) if is_device else _ptr(np.ascontiguousarray(src, dtype=np.float32))
        self._check(self._lib.mirt_load_accumulator(self._ctx, ptr, int(is_device), accumulations))

    def counters(self) -> dict:
        c = Counters()
        self._check(self._lib.mirt_get_counters(self._ctx, C.byref(c)))
        return c.as_dict()

    def kernel_times(self, reset: bool = True) -> dict:
        t = KernelTimes()
        self._check(self._lib.mirt_get_kernel_times(self._ctx, C.byref(t), int(reset)))
        return {k: {"ms": t.ms[i], "launches": int(t.launches[i])} for i, k in enumerate(KERNEL_CLASSES)}

    def debug_info(self) -> dict:
        out = (C.c_uint32 * 8)()
        self._check(self._lib.mirt_debug_info(self._ctx, out))
        keys = ("records", "lds_records", "lds_spheres", "depth", "half_boxes", "trace_lds_bytes", "trace_workgroups_per_cu", "cus")
        d = dict(zip(keys, [int(v) for v in out]))
        d["wide"] = (d["half_boxes"] >> 1) & 1             # 64-B binary16 records of up to four children
        d["half_boxes"] &= 1
        return d

    def debug_primary_lists(self) -> list:
        """hist[n] = pixels whose candidate list holds n spheres (n = 0..7), hist[8] = 8 or more, hist[9] = pixels without a list."""
        out = (C.c_uint32 * 10)()
        self._check(self._lib.mirt_debug_primary_lists(self._ctx, out))
        return [int(v) for v in out]

    def stream_handle(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.mirt_get_stream(self._ctx, C.byref(p)))
        return p.value

    # -- stage-level (parity tests) ---------------------------------------------------------------
    def debug_raygen(self, accumulations: int):
        n = (self.width // 16) * (self.height // 16) * 256
        p = np.empty((3, n), dtype=np.float32)
        d = np.empty((3, n), dtype=np.float32)
        self._check(self._lib.mirt_debug_raygen(self._ctx, accumulations, _ptr(p), _ptr(d)))
        return p, d

    def debug_trace_closest(self, p: np.ndarray, d: np.ndarray):
        p = np.ascontiguousarray(p, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32)
        n = p.shape[1]
        tfar = np.empty(n, dtype=np.float32); prim = np.empty(n, dtype=np.int32)
        self._check(self._lib.mirt_debug_trace_closest(self._ctx, n, _ptr(p), _ptr(d), _ptr(tfar), _ptr(prim)))
        return tfar, prim

    def debug_trace_shadow(self, p: np.ndarray, d: np.ndarray, tfar: np.ndarray):
        p = np.ascontiguousarray(p, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32)
        tfar = np.ascontiguousarray(tfar, dtype=np.float32)
        n = p.shape[1]
        occ = np.empty(n, dtype=np.uint8)
        self._check(self._lib.mirt_debug_trace_shadow(self._ctx, n, _ptr(p), _ptr(d), _ptr(tfar), _ptr(occ)))
        return occ

    def debug_math(self, fn: int, inputs: np.ndarray, n_out: int) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs, dtype=np.float32)
        n = inputs.shape[-1]
        out = np.empty((n_out, n), dtype=np.float32)
        self._check(self._lib.mirt_debug_math(self._ctx, fn, n, _ptr(inputs), _ptr(out)))
        return out


class GroupRenderer:
    """The same `Renderer<Policy>` interface on several GPUs of one node, driven by this one process through the library's
    mirt_group_* entry points (include/mirt.h): scene replicated, tile rows interleaved over the devices, one RCCL gather of the
    accumulator slabs to devices[0] when a frame or the accumulator is read.  `devices` may repeat a device (rehearsal on a
    one-GPU box; slabs then move with device copies)."""

    def __init__(self, scene: Scene, devices=(0,), max_bounces: int = 16, buckets: int = 5, mis: bool = True, use_bvh: bool = True,
                 count_traffic: bool = False, max_batch: int = 0, streams: int = 0, reference_tree: bool = False, gpu_build: bool = False):
        self._lib = load_library()
        self._g = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        rc = self._lib.mirt_group_create(dev, len(devices), C.byref(self._g))
        if rc != MIRT_OK:
            raise MirtError(f"mirt_group_create failed ({rc}): {self._lib.mirt_group_last_error(None).decode()}")
        self.scene, self.devices = scene, tuple(devices)
        self.width = self.height = 0
        self.framebuffer = None
        self.policy = Policy(max_bounces, buckets, int(mis), int(use_bvh), int(count_traffic), 0, max_batch, int(reference_tree), int(streams), int(gpu_build))
        self._check(self._lib.mirt_group_set_policy(self._g, C.byref(self.policy)))
        self.UpdateScene()

    def _check(self, rc):
        if rc < 0:
            raise MirtError(f"mirt_group call failed ({rc}): {self._lib.mirt_group_last_error(self._g).decode()}")
        return rc

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self._lib.mirt_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def UpdateScene(self):
        s = self.scene
        self.geometry = np.ascontiguousarray(s.geometry, dtype=SPHERE)
        self.material = np.ascontiguousarray(s.material, dtype=MATERIAL)
        self.nodes, self.prims = bvh_build(self.geometry)
        self.lights = light_list(self.geometry, self.material)
        hdri = np.ascontiguousarray(s.hdri, dtype=np.float32)
        amb = np.ascontiguousarray(s.ambient, dtype=np.float32)
        lights = self.lights if len(self.lights) else np.zeros(1, dtype=np.int32)
        self._check(self._lib.mirt_group_set_scene(self._g, _ptr(self.geometry), _ptr(self.prims), len(self.geometry), _ptr(self.nodes), len(self.nodes),
                                                   _ptr(self.material), len(self.material), _ptr(lights), len(self.lights), _ptr(amb),
                                                   _ptr(hdri), hdri.shape[1], hdri.shape[0]))
        self.UpdateCamera()

    def UpdateCamera(self):
        cam: Camera = self.scene.camera
        pos = np.ascontiguousarray(cam.pos, dtype=np.float32)
        ori = np.ascontiguousarray(cam.orient, dtype=np.float32)
        self._check(self._lib.mirt_group_set_camera(self._g, _ptr(pos), _ptr(ori), float(cam.half_width), float(cam.half_height), float(cam.z), float(cam.exposure)))

    def Resize(self, new_width: int, new_height: int):
        self.width, self.height = int(new_width), int(new_height)
        self.scene.camera.resize(self.width, self.height)
        self.UpdateCamera()
        self._check(self._lib.mirt_group_resize(self._g, self.width, self.height))
        self.framebuffer = np.zeros((self.height, self.width, 4), dtype=np.float32)

    def ResetAccumulator(self):
        self._check(self._lib.mirt_group_reset(self._g))

    def Accumulate(self, n_calls: int = 1):
        self._check(self._lib.mirt_group_accumulate(self._g, n_calls))

    def AccumulateAsync(self, n_calls: int = 1):
        self._check(self._lib.mirt_group_accumulate_async(self._g, n_calls))

    def Synchronize(self):
        self._check(self._lib.mirt_group_synchronize(self._g))

    def Render(self) -> bool:
        return self._check(self._lib.mirt_group_render(self._g, _ptr(self.framebuffer))) == MIRT_OK

    def GetFrame(self) -> np.ndarray:
        return self.framebuffer

    @property
    def accumulations(self) -> int:
        v = C.c_uint32(0)
        self._check(self._lib.mirt_group_get_accumulations(self._g, C.byref(v)))
        return v.value

    def accumulator(self) -> np.ndarray:
        """The whole image's [tile][bucket][rgb][256] slab in LaunchIndex order (gathers first)."""
        n = C.c_size_t(0)
        self._check(self._lib.mirt_group_accumulator_floats(self._g, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self._lib.mirt_group_read_accumulator(self._g, _ptr(out)))
        return out.reshape(n.value // (self.policy.buckets * 768), self.policy.buckets, 3, 256)

    def counters(self) -> dict:
        c = Counters()
        self._check(self._lib.mirt_group_get_counters(self._g, C.byref(c)))
        return c.as_dict()

    def Gather(self):
        """The one exchange of the path: every member's accumulator slab to devices[0] (implied by accumulator() and Render())."""
        self._check(self._lib.mirt_group_gather(self._g))

    def gather_ms(self) -> float:
        v = C.c_double(0.0)
        self._check(self._lib.mirt_group_last_gather_ms(self._g, C.byref(v)))
        return v.value


def rccl_selftest(device: int = 0, n_floats: int = 1 << 20) -> None:
    """mirt_group_rccl_selftest: raises if librccl cannot be loaded or a grouped ncclSend / ncclRecv on `device` does not deliver."""
    lib = load_library()
    rc = lib.mirt_group_rccl_selftest(device, n_floats)
    if rc != MIRT_OK:
        raise MirtError(f"RCCL self-test failed ({rc}): {lib.mirt_group_last_error(None).decode()}")
