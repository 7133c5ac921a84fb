"""Scene authoring for the headless harness (stands in for RaytracingApp's constructor,
Application.cpp:32-235): the reference's scene structs as numpy record arrays in their exact byte
layouts, its Camera/Projection/View host math, the two usable built-in scenes, and the synthetic
S(n) benchmark scenes of SURVEY.md §8d.

Everything here is *input* to the hot path (both the HIP path and the test oracle receive the same
arrays); none of it is on the path itself.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

f32 = np.float32

# Primitives.hpp:7-17 (32 B), :18-27 (96 B); BVH.hpp:18-31 (32 B)
SPHERE = np.dtype([("position", f32, 3), ("radius_sq", f32), ("material_ID", np.int32), ("_pad", np.int32, 3)])
MATERIAL = np.dtype([("albedo", f32, 3), ("F0", f32, 3), ("F80", f32, 3), ("emission", f32, 3),
                     ("transmission", f32, 3), ("roughness", f32), ("IOR_minus_one", f32), ("_pad", f32, 7)])
NODE = np.dtype([("min_bound", f32, 3), ("first_id", np.uint32), ("max_bound", f32, 3), ("prim_count", np.uint32)])
assert SPHERE.itemsize == 32 and MATERIAL.itemsize == 96 and NODE.itemsize == 32


# ----------------------------------------------------------------------------------------------
# Random.hpp:5-43 — used here only as the deterministic generator for synthetic scenes
# ----------------------------------------------------------------------------------------------
def hash_u32(i: int) -> int:
    i &= 0xFFFFFFFF
    i ^= i >> 16
    i = (i * 0x21F0AAAD) & 0xFFFFFFFF
    i ^= i >> 15
    i = (i * 0xD35A2D97) & 0xFFFFFFFF
    i ^= i >> 15
    return i ^ 0xE6FE3BEB


def pcg_floats(state: int, count: int) -> np.ndarray:
    """`count` successive rand_unit_float draws (Random.hpp:20-29) as float32."""
    out = np.empty(count, dtype=np.uint32)
    s = state & 0xFFFFFFFF
    for k in range(count):
        v = s
        s = (s * 747796405 + 2891336453) & 0xFFFFFFFF
        v = (((v >> ((v >> 28) + 4)) ^ v) * 277803737) & 0xFFFFFFFF
        out[k] = (v >> 22) ^ v
    return out.astype(f32) * f32(2.0 ** -32)


# ----------------------------------------------------------------------------------------------
# Camera.hpp — host-side setup (runs once per camera change, not on the hot path)
# ----------------------------------------------------------------------------------------------
def _normalize(v):
    v = np.asarray(v, dtype=f32)
    return v * (f32(1.0) / np.sqrt(np.dot(v, v), dtype=f32))


def quat_look_at(direction, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """glm::quatLookAt (RH) -> quaternion stored (x, y, z, w); View ctor, Camera.hpp:48-50."""
    d = np.asarray(direction, dtype=f32)
    c2 = -d
    right = np.cross(np.asarray(up, dtype=f32), c2).astype(f32)
    c0 = right * (f32(1.0) / np.sqrt(max(f32(0.00001), np.dot(right, right)), dtype=f32))
    c1 = np.cross(c2, c0).astype(f32)
    m = [c0, c1, c2]  # m[col][row]
    four_x = m[0][0] - m[1][1] - m[2][2]
    four_y = m[1][1] - m[0][0] - m[2][2]
    four_z = m[2][2] - m[0][0] - m[1][1]
    four_w = m[0][0] + m[1][1] + m[2][2]
    biggest, idx = four_w, 0
    for i, v in ((1, four_x), (2, four_y), (3, four_z)):
        if v > biggest:
            biggest, idx = v, i
    big = np.sqrt(f32(biggest) + f32(1.0), dtype=f32) * f32(0.5)
    mult = f32(0.25) / big
    if idx == 0:
        w, x, y, z = big, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult
    elif idx == 1:
        w, x, y, z = (m[1][2] - m[2][1]) * mult, big, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult
    elif idx == 2:
        w, x, y, z = (m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, big, (m[1][2] + m[2][1]) * mult
    else:
        w, x, y, z = (m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, big
    return np.array([x, y, z, w], dtype=f32)


@dataclass
class Camera:
    """Camera / View / Projection fields the hot path reads (Camera.hpp:61-88)."""
    eye: tuple = (0.0, 0.0, 0.0)
    direction: tuple = (0.0, 0.0, -1.0)
    focal_length: float = 50.0
    exposure: float = 1.0
    pos: np.ndarray = field(init=False)
    orient: np.ndarray = field(init=False)     # x, y, z, w
    half_width: np.float32 = field(init=False, default=f32(0.5))
    half_height: np.float32 = field(init=False, default=f32(0.5))
    z: np.float32 = field(init=False, default=f32(0))

    def __post_init__(self):
        self.pos = np.asarray(self.eye, dtype=f32)
        self.orient = quat_look_at(_normalize(self.direction))
        self.resize(1, 1)

    def resize(self, width: int, height: int):
        """Projection::Resize + UpdateLens (Camera.hpp:20-31): z = half_height * ((-2/24) * focal_length)."""
        self.half_height = f32(height) * f32(0.5)
        self.half_width = f32(width) * f32(0.5)
        inv_half_tan = (f32(-2.0) / f32(24.0)) * f32(self.focal_length)
        self.z = self.half_height * inv_half_tan


@dataclass
class Scene:
    """Scene aggregate (Scene.hpp:19-26) before acceleration structures are built."""
    geometry: np.ndarray
    material: np.ndarray
    camera: Camera
    ambient: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=f32))
    hdri: np.ndarray = field(default_factory=lambda: np.ones((1, 1, 4), dtype=f32))   # synthetic 1x1 texel (env.hdr is not in the repo)
    name: str = "scene"


def _sphere(pos, radius_sq, mat):
    s = np.zeros((), dtype=SPHERE)
    s["position"] = np.asarray(pos, dtype=f32)
    s["radius_sq"] = f32(radius_sq)
    s["material_ID"] = mat
    return s


def _material(albedo=(0, 0, 0), emission=(0, 0, 0)):
    m = np.zeros((), dtype=MATERIAL)
    m["albedo"] = np.asarray(albedo, dtype=f32)
    m["emission"] = np.asarray(emission, dtype=f32)
    return m


def default9() -> Scene:
    """Scenes::Default, Application.cpp:33-101 (9 spheres / 9 materials, 3 emissive, ambient 0)."""
    f = f32
    mats = [
        _material(albedo=(1, 1, 1)),
        _material(albedo=(1, 1, 1), emission=f(0.1) * np.array([25.0, 25.0, 200.0], dtype=f32)),
        _material(albedo=(1, 1, 1), emission=f(0.1) * np.array([150.0, 150.0, 150.0], dtype=f32)),
        _material(albedo=(1, 1, 1), emission=(200.0, 17.0, 25.0)),
        _material(albedo=(0.793, 0.793, 0.664)),
        _material(albedo=(0.05, 0.05, 0.05)),
        _material(albedo=(1, 1, 1)),
        _material(albedo=(1, 1, 1)),
        _material(albedo=(1, 1, 1)),
    ]
    geo = [
        _sphere((0.3, -1.47, 0.0), f(1.5) * f(1.5), 0),
        _sphere((0.29999, 0.0801, 0.0), f(0.05) * f(0.05), 1),
        _sphere((0.3302, 0.36165, 0.7119), f(0.05) * f(0.05), 2),
        _sphere((-0.4857, -0.0242, -0.41383), f(0.05) * f(0.05), 3),
        _sphere((0.3, 1.7, 0.0), f(1.5) * f(1.5), 4),
        _sphere((0.018, 0.022, 0.07), f(0.02) * f(0.02), 5),
        _sphere((-0.037, 0.022, 0.00), f(0.03) * f(0.03), 6),
        _sphere((-0.0846, -0.0334, 0.283), f(0.012) * f(0.012), 7),
        _sphere((0.03863, -0.00788, 0.2835), f(0.012) * f(0.012), 8),
    ]
    cam = Camera(eye=(-0.2, 0.3, 1.0), direction=(0.1, -0.4, -1.0), focal_length=40.0, exposure=1.0)
    return Scene(np.array(geo, dtype=SPHERE), np.array(mats, dtype=MATERIAL), cam, np.zeros(3, dtype=f32), name="default9")


def white_furnace() -> Scene:
    """Scenes::White_Furnace, Application.cpp:218-223: albedo-1 unit sphere under a constant-1 sky.
    No emissive sphere, so NEE is skipped by the light_count == 0 guard."""
    cam = Camera(eye=(0, 0, 3), direction=(0, 0, -1), focal_length=50.0, exposure=1.0)
    return Scene(np.array([_sphere((0, 0, 0), 1.0, 0)], dtype=SPHERE), np.array([_material(albedo=(1, 1, 1))], dtype=MATERIAL),
                 cam, np.ones(3, dtype=f32), name="white_furnace")


def pcg_stream(state: int):
    """Generator of successive pcg_generate outputs (Random.hpp:20-24) as Python ints."""
    s = state & 0xFFFFFFFF
    while True:
        v = s
        s = (s * 747796405 + 2891336453) & 0xFFFFFFFF
        v = (((v >> ((v >> 28) + 4)) ^ v) * 277803737) & 0xFFFFFFFF
        yield (v >> 22) ^ v


def bvh_test() -> Scene:
    """Scenes::BVH_test, Application.cpp:102-122, in a FIXED version: 255 random spheres (radius 0.3..20, x/z in +-100, y in 0..100)
    under a constant-1 sky, camera {0,60,300} looking down -z.  As shipped the scene cannot run: its material list is empty
    (`mat_dist(0, size-1)` with size 0) and std::mt19937 + std::uniform_*_distribution differ between standard libraries.
    Fixed here (and identically in csrc/mirt_headless.cpp): eight Lambertian materials, and the reference's own PCG
    (Random.hpp:20-34) seeded with hash_u32 of the low 32 bits of the shipped seed; draws per sphere in the shipped order:
    radius, x, y, z, material."""
    g = pcg_stream(hash_u32(0x04D15A07))
    unit = lambda: f32(next(g)) * f32(2.0 ** -32)       # noqa: E731  rand_unit_float
    mats = np.zeros(8, dtype=MATERIAL)
    for m in range(8):
        for c in range(3):
            mats["albedo"][m, c] = f32(0.2) + f32(0.7) * unit()
    geo = np.zeros(255, dtype=SPHERE)
    for i in range(255):
        r = f32(0.3) + f32(19.7) * unit()
        x = f32(-100.0) + f32(200.0) * unit()
        y = f32(100.0) * unit()
        z = f32(-100.0) + f32(200.0) * unit()
        m = min(7, int(np.uint32(unit() * f32(8.0))))    # rand_bounded_int(state, 8)
        geo[i] = _sphere((x, y, z), r * r, m)
    cam = Camera(eye=(0.0, 60.0, 300.0), direction=(0.0, 0.0, -1.0), focal_length=50.0, exposure=1.0)
    return Scene(geo, mats, cam, np.ones(3, dtype=f32), name="bvh_test")


def synthetic(n: int, ambient: float = 0.0, scene_seed: int = 1) -> Scene:
    """S(n) of SURVEY.md §8d: ground sphere + n-1 random spheres at constant density, 16 Lambertian
    materials + one emissive material on every 64th sphere; generator = the reference's own PCG."""
    assert n >= 2
    cbrt = f32(np.cbrt(np.float64(n)))
    H, L = cbrt, cbrt * f32(2.0)
    u = pcg_floats(hash_u32(scene_seed), 16 * 3 + (n - 1) * 4)
    mats = np.zeros(17, dtype=MATERIAL)
    mats["albedo"][:16] = (f32(0.2) + f32(0.6) * u[:48]).reshape(16, 3)
    mats["albedo"][16] = 1.0
    mats["emission"][16] = 20.0
    r = u[48:].reshape(n - 1, 4)
    geo = np.zeros(n, dtype=SPHERE)
    geo["position"][0] = (0.0, -1000.0, 0.0)
    geo["radius_sq"][0] = f32(1000.0) * f32(1000.0)
    geo["material_ID"][0] = 0
    geo["position"][1:, 0] = -L + (f32(2.0) * L) * r[:, 0]
    geo["position"][1:, 1] = f32(0.2) + (H - f32(0.2)) * r[:, 1]
    geo["position"][1:, 2] = -L + (f32(2.0) * L) * r[:, 2]
    rad = f32(0.3) + f32(0.7) * r[:, 3]
    geo["radius_sq"][1:] = rad * rad
    idx = np.arange(n)
    geo["material_ID"] = (idx % 16).astype(np.int32)
    emissive = (idx % 64) == 63
    if n < 64:
        emissive[n - 1] = True
    geo["material_ID"][emissive] = 16
    cam = Camera(eye=(0.0, float(H), float(f32(3.0) * L)), direction=(0.0, -0.3, -1.0), focal_length=40.0, exposure=1.0)
    amb = np.full(3, ambient, dtype=f32)
    return Scene(geo, mats, cam, amb, name=f"S({n})")


# BASELINE.json configs: width, height, accumulations, Policy.max_bounces, spheres, use_bvh, buckets
CONFIGS = {
    "cfg1": dict(width=512, height=512, spp=1, max_bounces=2, n=8, use_bvh=0, buckets=5, ambient=0.5),
    "cfg2": dict(width=1024, height=1024, spp=64, max_bounces=5, n=1000, use_bvh=1, buckets=5, ambient=0.5),
    "cfg3": dict(width=1920, height=1088, spp=256, max_bounces=9, n=10000, use_bvh=1, buckets=5, ambient=0.0),
    "cfg4": dict(width=4096, height=4096, spp=256, max_bounces=9, n=100000, use_bvh=1, buckets=5, ambient=0.0),
    "cfg5": dict(width=4096, height=4096, spp=1024, max_bounces=17, n=100000, use_bvh=1, buckets=16, ambient=0.0),
}
