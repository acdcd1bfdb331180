"""Known-answer tests that pin the CPU oracle (oracle/).  PARITY UNPINNED against real DXR output: the
reference ships no tests or golden vectors for this path; the only reference-supplied
fixture is the Sheen_E LUT.  Everything else here is analytic: inverses, normalisations, Monte-Carlo
consistency of Sample*/Pdf* pairs, and vectors computed from the published definitions."""
import ctypes as C
import math

import numpy as np
import pytest

from gltf_renderer_amd import abi


def vec(oracle_lib, name, inp, n_out, *pre):
    return oracle_lib.call_vec(name, inp, n_out, *pre)


# ---- Random.hlsli:17-30 ------------------------------------------------------------------------
def pcg4d_py(v):
    M = 0xFFFFFFFF
    v = [(x * 1664525 + 1013904223) & M for x in v]
    v[0] = (v[0] + v[1] * v[3]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M; v[3] = (v[3] + v[1] * v[2]) & M
    v = [x ^ (x >> 16) for x in v]
    v[0] = (v[0] + v[1] * v[3]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M; v[3] = (v[3] + v[1] * v[2]) & M
    return v


def test_pcg4d_matches_definition(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        v = rng.integers(0, 2 ** 32, 4, dtype=np.uint64).astype(np.uint32)
        out = np.zeros(4, np.uint32)
        L.orc_pcg4d(v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert list(out) == pcg4d_py([int(x) for x in v])
    # fixed vectors (computed from the 10-line definition)
    assert pcg4d_py([0, 0, 0, 0]) == [int(x) for x in _pcg(L, [0, 0, 0, 0])]
    assert pcg4d_py([1, 2, 3, 4]) == [int(x) for x in _pcg(L, [1, 2, 3, 4])]


def _pcg(L, v):
    a = np.array(v, np.uint32); out = np.zeros(4, np.uint32)
    L.orc_pcg4d(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def test_random_float_range_and_quirk_q29(oracle_lib):
    L = oracle_lib.lib()
    out = np.zeros(4, np.float32)
    vals = []
    for c in range(500):
        L.orc_random(17, 33, 5, c, out.ctypes.data_as(C.c_void_p))
        r = pcg4d_py([17, 33, 5, c])
        # uint -> float (RNE) then fp32 divide by the fp32 value of 4294967295.0 = 2^32
        expect = (np.array(r, np.uint32).astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
        assert np.array_equal(out, expect)
        vals += list(out)
    assert 0.0 <= min(vals) and max(vals) <= 1.0
    # u == 1.0 is reachable: r >= 0xFFFFFF80 rounds to 2^32
    assert np.float32(np.uint32(0xFFFFFF80)) / np.float32(4294967296.0) == np.float32(1.0)


# ---- octahedral / tangent space ------------------------------------------------------------------
def test_octahedral_round_trip(oracle_lib):
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        e = vec(oracle_lib, "orc_octa_encode", n, 2)
        assert np.all(np.abs(e) <= 1 + 1e-6)
        d = vec(oracle_lib, "orc_octa_decode", e, 3)
        assert np.allclose(d, n, atol=2e-6)


def test_tangent_space_pack_unpack_negates_tangent_quirk_q25(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        t = np.cross(n, rng.normal(size=3)); t /= np.linalg.norm(t)
        w = 1.0 if rng.random() < 0.5 else -1.0
        nf = np.array(n, np.float32); tf = np.array([*t, w], np.float32)
        for enc in (L.orc_encode_tangent_space_host, L.orc_encode_tangent_space_shader):
            enc.restype = C.c_uint32
            packed = enc(nf.ctypes.data_as(C.c_void_p), tf.ctypes.data_as(C.c_void_p))
            dn, dt = np.zeros(3, np.float32), np.zeros(4, np.float32)
            L.orc_decode_tangent_space(packed, dn.ctypes.data_as(C.c_void_p), dt.ctypes.data_as(C.c_void_p))
            assert np.dot(dn, n) > 0.9999
            # encode stores angle/2pi + 0.5, decode uses 2pi*z without removing the 0.5 -> tangent negated
            assert np.dot(dt[:3], t) < -0.999
            assert dt[3] == w
            assert abs(np.dot(dt[:3], dn)) < 2e-3


def test_encode_normal_host_has_zero_angle_bits(oracle_lib):
    L = oracle_lib.lib()
    n = np.array([0.3, -0.5, 0.81], np.float32); n /= np.linalg.norm(n)
    p = L.orc_encode_normal_host(n.ctypes.data_as(C.c_void_p))
    assert (p >> 20) & 0x3ff == 0 and (p >> 30) == 3


def test_numpy_packer_matches_oracle_encoder(oracle_lib):
    """Host-logic packer (gltf_renderer_amd.meshgen, Gltf.cpp:79-104) vs the oracle's C++ restatement."""
    from gltf_renderer_amd import meshgen
    L = oracle_lib.lib()
    rng = np.random.default_rng(3)
    n = rng.normal(size=(2000, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    t = np.cross(n, rng.normal(size=(2000, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)
    t4 = np.concatenate([t, np.where(rng.random((2000, 1)) < 0.5, 1.0, -1.0)], axis=1).astype(np.float32)
    nf = n.astype(np.float32)
    mine = meshgen.encode_tangent_space(nf, t4)
    same = 0
    for i in range(2000):
        ref = L.orc_encode_tangent_space_host(nf[i].ctypes.data_as(C.c_void_p), t4[i].ctypes.data_as(C.c_void_p))
        if ref == int(mine[i]):
            same += 1
        else:   # at most one quantisation step apart in the angle field (atan2 last-bit differences)
            assert (ref & 0xC00FFFFF) == (int(mine[i]) & 0xC00FFFFF)
            assert abs(((ref >> 20) & 0x3ff) - ((int(mine[i]) >> 20) & 0x3ff)) <= 1
    assert same >= 1990


# ---- Transforms.hlsli ---------------------------------------------------------------------------------
def test_square_sphere_inverse_and_unit_length(oracle_lib):
    rng = np.random.default_rng(4)
    for _ in range(500):
        s = rng.uniform(-1, 1, 2)
        d = vec(oracle_lib, "orc_square_to_sphere", s, 3)
        assert abs(np.linalg.norm(d) - 1) < 2e-6
        s2 = vec(oracle_lib, "orc_sphere_to_square", d, 2)
        assert np.allclose(s2, s, atol=3e-5)


def test_square_to_sphere_is_equal_area(oracle_lib):
    rng = np.random.default_rng(5)
    n = 40000
    z = np.array([vec(oracle_lib, "orc_square_to_sphere", rng.uniform(-1, 1, 2), 3)[2] for _ in range(n)])
    hist, _ = np.histogram(z, bins=10, range=(-1, 1))       # uniform on the sphere <=> z uniform in [-1, 1]
    assert np.all(np.abs(hist / n - 0.1) < 0.01)


def test_uv_square_inverse(oracle_lib):
    for uv in ([0.1, 0.7], [0.0, 0.0], [1.0, 1.0], [0.33, 0.5]):
        s = vec(oracle_lib, "orc_uv_to_square", uv, 2)
        assert np.allclose(vec(oracle_lib, "orc_square_to_uv", s, 2), uv, atol=1e-6)
    assert np.allclose(vec(oracle_lib, "orc_uv_to_square", [0, 0], 2), [-1, 1])       # v flips


def test_square_to_disk_inside_unit_disk_and_sign0(oracle_lib):
    rng = np.random.default_rng(6)
    for _ in range(200):
        d = vec(oracle_lib, "orc_square_to_disk", rng.uniform(-1, 1, 2), 2)
        assert np.linalg.norm(d) <= 1 + 1e-6
    assert np.allclose(vec(oracle_lib, "orc_square_to_disk", [0, 0], 2), [0, 0])


def test_cubemap_direction_face_round_trip(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(7)
    for face in range(6):
        for _ in range(50):
            uv = rng.uniform(0.01, 0.99, 2).astype(np.float32)
            d = np.zeros(3, np.float32)
            L.orc_cubemap_to_direction(face, uv.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
            f = C.c_int(); uv2 = np.zeros(2, np.float32)
            L.orc_dir_to_face(d.ctypes.data_as(C.c_void_p), C.byref(f), uv2.ctypes.data_as(C.c_void_p))
            assert f.value == face and np.allclose(uv2, uv, atol=2e-6)


# ---- fixtures / conversions ---------------------------------------------------------------------------
def test_sheen_lut_fixture_pinned(oracle_lib):
    """The one reference-supplied fixture: Resources/Sheen_E.exr decoded (tests/golden/sheen_e_16x16.npy)."""
    lut = oracle_lib.sheen_lut()
    assert lut.shape == (16, 16)
    assert abs(lut.min() - 2.3841858e-06) < 1e-12 and abs(lut.max() - 0.8730469) < 1e-7
    assert abs(lut[0, 0] - 0.61279297) < 1e-7 and abs(lut[15, 0] - 0.8730469) < 1e-7
    o = oracle_lib.Oracle()
    L = oracle_lib.lib()
    for j in (0, 5, 15):
        for i in (0, 7, 15):       # at texel centres the bilinear LUT lookup returns the texel (row = alpha, col = cos_theta)
            assert abs(L.orc_sheen_e(o.h, (j + 0.5) / 16, (i + 0.5) / 16) - lut[j, i]) < 1e-7
    assert abs(L.orc_sheen_e(o.h, 0.0, 0.0) - lut[0, 0]) < 1e-7             # clamp addressing
    mid = L.orc_sheen_e(o.h, 0.5 / 16, 1.0 / 16)                            # halfway between texel 0 and 1 in u
    assert abs(mid - 0.5 * (lut[0, 0] + lut[0, 1])) < 1e-6
    # the product's shipped table is the same data
    from gltf_renderer_amd.renderer import sheen_lut
    assert np.array_equal(sheen_lut().reshape(16, 16), lut)


def test_half_conversions_match_ieee(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(8)
    vals = np.concatenate([rng.normal(size=500) * 10 ** rng.uniform(-8, 5, 500), [0, 1, 65504, 65519.9, 65520, 1e9, 6e-8, 5.9e-8, 2.98e-8, -3.5]]).astype(np.float32)
    for v in vals:
        h = L.orc_float_to_half(float(v))
        expect = np.float32(v).astype(np.float16)
        assert h == int(expect.view(np.uint16)), (v, h, expect.view(np.uint16))
        assert L.orc_half_to_float(h) == np.float32(expect) or (np.isinf(expect) and np.isinf(L.orc_half_to_float(h)))


def test_mat4_inverse(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(9)
    for _ in range(20):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        flat = np.ascontiguousarray(m.T.reshape(16))
        out = np.zeros(16, np.float32)
        L.orc_mat4_inverse(flat.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.allclose(out.reshape(4, 4).T, np.linalg.inv(m.astype(np.float64)), rtol=1e-4, atol=1e-5)


def test_offset_ray_moves_to_normal_side(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(10)
    for scale in (1e-3, 0.02, 1.0, 50.0, 900.0):
        for _ in range(50):
            p = (rng.normal(size=3) * scale).astype(np.float32)
            n = rng.normal(size=3); n = (n / np.linalg.norm(n)).astype(np.float32)
            out = np.zeros(3, np.float32)
            L.orc_offset_ray(p.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            d = out.astype(np.float64) - p
            assert np.dot(d, n) > 0
            assert np.linalg.norm(d) < max(1e-3 * scale, 1e-4)
    # small-coordinate branch: |p| < 1/32 uses p + n/65536
    p = np.array([0.01, -0.02, 0.0], np.float32); n = np.array([0, 0, 1], np.float32); out = np.zeros(3, np.float32)
    L.orc_offset_ray(p.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert out[2] == np.float32(1.0 / 65536.0) and out[0] == p[0]


# ---- lights (Lights.hlsli:26-61) ---------------------------------------------------------------------------
def _light_ray(oracle_lib, light, p):
    L = oracle_lib.lib()
    pp = np.array(p, np.float32); out = np.zeros(6, np.float32)
    L.orc_light_ray(C.byref(light), pp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out[:3], out[3:]


def test_light_rays(oracle_lib):
    l = abi.PtLight(); l.type = abi.LIGHT_POINT; l.position[:] = (0, 0, 2); l.color[:] = (1, 0.5, 0.25); l.intensity = 8.0; l.cutoff = 0.0
    d, c = _light_ray(oracle_lib, l, (0, 0, 0))
    assert np.allclose(d, (0, 0, 1)) and np.allclose(c, np.array([1, 0.5, 0.25]) * 8 / 4)          # inverse square
    l.cutoff = 4.0
    d, c = _light_ray(oracle_lib, l, (0, 0, 0))
    assert np.allclose(c, np.array([1, 0.5, 0.25]) * 8 / 4 * (1 - (2 / 4) ** 4), rtol=1e-6)        # range window
    d, c = _light_ray(oracle_lib, l, (0, 0, -3))
    assert np.allclose(c, 0)                                                                        # beyond the range
    l.type = abi.LIGHT_DIRECTIONAL; l.direction[:] = (0, 0, -1)
    d, c = _light_ray(oracle_lib, l, (5, 5, 5))
    assert np.allclose(d, (0, 0, 1)) and np.allclose(c, np.array([1, 0.5, 0.25]) * 8)
    l.type = abi.LIGHT_SPOT; l.cutoff = 0; l.position[:] = (0, 0, 2); l.direction[:] = (0, 0, -1); l.inner_angle = 0.2; l.outer_angle = 0.5
    d, c_axis = _light_ray(oracle_lib, l, (0, 0, 0))
    assert np.allclose(c_axis, np.array([1, 0.5, 0.25]) * 8 / 4, rtol=1e-6)                         # inside the inner cone
    d, c_out = _light_ray(oracle_lib, l, (2 * math.tan(0.6), 0, 0))
    assert np.allclose(c_out, 0)                                                                    # outside the outer cone
    ang = 0.35
    d, c_mid = _light_ray(oracle_lib, l, (2 * math.tan(ang), 0, 0))
    dist2 = 4 + (2 * math.tan(ang)) ** 2
    sm = (math.cos(ang) - math.cos(0.5)) / (math.cos(0.2) - math.cos(0.5))
    assert np.allclose(c_mid, np.array([1, 0.5, 0.25]) * 8 / dist2 * sm * sm, rtol=1e-4)


# ---- software texture unit ---------------------------------------------------------------------------------
def test_texture_sampling_rules(oracle_lib):
    o = oracle_lib.Oracle()
    L = oracle_lib.lib()
    ramp = np.zeros((2, 4, 4), np.uint8)
    ramp[..., 0] = np.array([0, 85, 170, 255])[None, :]
    ramp[1, :, 1] = 255
    ramp[..., 3] = 255
    t_lin = o.texture_create(ramp, False)
    t_srgb = o.texture_create(ramp, True)
    s_clamp = o.sampler_create(abi.ADDRESS_CLAMP, abi.ADDRESS_CLAMP, abi.FILTER_LINEAR, abi.FILTER_LINEAR)
    s_mirror = o.sampler_create(abi.ADDRESS_MIRROR, abi.ADDRESS_MIRROR, abi.FILTER_LINEAR, abi.FILTER_LINEAR)
    s_point = o.sampler_create(abi.ADDRESS_WRAP, abi.ADDRESS_WRAP, abi.FILTER_POINT, abi.FILTER_POINT)

    def tap(tex, smp, u, v):
        uv = np.array([u, v], np.float32); out = np.zeros(4, np.float32)
        L.orc_sample_texture(o.h, tex, smp, uv.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out
    # texel centres at (i + 0.5) / N return the texel
    for i, r in enumerate([0, 85, 170, 255]):
        assert abs(tap(t_lin, 0, (i + 0.5) / 4, 0.25)[0] - r / 255) < 1e-6
    # halfway between texel 1 and 2
    assert abs(tap(t_lin, 0, 0.5, 0.25)[0] - 0.5 * (85 + 170) / 255) < 1e-6
    # wrap: u = 0 blends texel 3 and 0; clamp: returns texel 0; mirror: texel 0 with itself
    assert abs(tap(t_lin, 0, 0.0, 0.25)[0] - 0.5) < 1e-6
    assert abs(tap(t_lin, s_clamp, 0.0, 0.25)[0] - 0.0) < 1e-6
    assert abs(tap(t_lin, s_mirror, 0.0, 0.25)[0] - 0.0) < 1e-6
    assert abs(tap(t_lin, s_mirror, 1.125, 0.25)[0] - 1.0) < 1e-6          # mirrored: texel 3 again
    assert abs(tap(t_lin, s_clamp, 7.0, 0.25)[0] - 1.0) < 1e-6
    # point filter picks floor(u * N)
    assert abs(tap(t_lin, s_point, 0.49, 0.25)[0] - 85 / 255) < 1e-6
    assert abs(tap(t_lin, s_point, 1.3, 0.25)[0] - 85 / 255) < 1e-6        # wraps
    # vertical bilinear between rows (green 0 -> 1)
    assert abs(tap(t_lin, s_clamp, 0.125, 0.5)[1] - 0.5) < 1e-6
    # sRGB is decoded BEFORE filtering; alpha is linear
    c = 170 / 255
    assert abs(tap(t_srgb, 0, 2.5 / 4, 0.25)[0] - ((c + 0.055) / 1.055) ** 2.4) < 1e-6
    a, b = ((85 / 255 + 0.055) / 1.055) ** 2.4, ((170 / 255 + 0.055) / 1.055) ** 2.4
    assert abs(tap(t_srgb, 0, 0.5, 0.25)[0] - 0.5 * (a + b)) < 1e-6
    assert tap(t_srgb, 0, 0.5, 0.25)[3] == 1.0


# ---- environment map ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def env_oracle(oracle_lib):
    from gltf_renderer_amd import scenes
    o = oracle_lib.Oracle()
    img = scenes.sky_image(256, 128, 200.0)
    e = o.env_create(img)
    return o, e, img


def test_environment_sizes_and_pyramid(oracle_lib, env_oracle):
    o, e, img = env_oracle
    n, cube, pyr = o.env_read(e)
    assert n == max((256 // 4) // 2, 1) + 1 == 33                           # EnvironmentMap.cpp:92 (quirk q11)
    lv = []
    off = 0
    for r in [1024 >> i for i in range(11)]:
        lv.append(pyr[off:off + r * r].reshape(r, r)); off += r * r
    assert np.all(lv[0] >= 0)
    for i in range(1, 11):                                                  # 2x2 SUM pyramid
        up = lv[i - 1]
        s = up[0::2, 0::2] + up[1::2, 0::2] + up[0::2, 1::2] + up[1::2, 1::2]
        assert np.allclose(lv[i], s, rtol=1e-6)
    assert abs(lv[10][0, 0] - lv[0].astype(np.float64).sum()) / lv[0].sum() < 1e-4


def test_importance_map_pdf_integrates_to_one_and_matches_sampler(oracle_lib, env_oracle):
    o, e, _ = env_oracle
    L = oracle_lib.lib()
    n, cube, pyr = o.env_read(e)
    lv0 = pyr[:1024 * 1024].reshape(1024, 1024).astype(np.float64)
    # pdf(texel) = W*H*texel/total  ->  mean over texels = 1
    assert abs((1024 * 1024 * lv0 / lv0.sum()).mean() - 1.0) < 1e-9
    rng = np.random.default_rng(11)
    counts = np.zeros((8, 8))
    N = 40000
    out = np.zeros(3, np.float32)
    for _ in range(N):
        u = rng.random(2).astype(np.float32)
        L.orc_sample_importance_map(o.h, e, u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        x, y = min(int(out[0] * 1024), 1023), min(int(out[1] * 1024), 1023)
        # returned pdf = pdf of the chosen texel ((pixel + u)/1024 can round onto the next texel when u is ~1)
        cands = [1024 * 1024 * lv0[max(y - dy, 0), max(x - dx, 0)] / lv0.sum() for dy in (0, 1) for dx in (0, 1)]
        assert min(abs(out[2] - c) for c in cands) <= 2e-3 * out[2] + 1e-6
        counts[y // 128, x // 128] += 1
    expect = lv0.reshape(8, 128, 8, 128).sum(axis=(1, 3)) / lv0.sum() * N
    big = expect > 200
    assert np.all(np.abs(counts[big] - expect[big]) < 5 * np.sqrt(expect[big]) + 0.02 * expect[big])


def test_importance_map_pdf_lookup_is_off_by_one_quirk_q9(oracle_lib, env_oracle):
    o, e, _ = env_oracle
    L = oracle_lib.lib()
    n, cube, pyr = o.env_read(e)
    lv0 = pyr[:1024 * 1024].reshape(1024, 1024)
    total = pyr[-1]
    for (x, y) in ((10, 20), (500, 700), (1, 1), (0, 0)):
        uv = np.array([(x + 0.5) / 1024, (y + 0.5) / 1024], np.float32)
        got = L.orc_importance_map_pdf(o.h, e, uv.ctypes.data_as(C.c_void_p))
        xs, ys = max(x - 1, 0), max(y - 1, 0)                             # UVToPixel = (int)(floor(uv*res) - 0.5)
        assert abs(got - 1024 * 1024 * lv0[ys, xs] / total) <= 1e-5 * abs(got) + 1e-9


def test_cube_lookup_follows_equal_area_equirect_quirk_q8(oracle_lib):
    """A sky that depends only on z must come back as that function of z (v = 1 - (z+1)/2)."""
    o = oracle_lib.Oracle()
    L = oracle_lib.lib()
    H, W = 256, 512
    v = (np.arange(H) + 0.5) / H
    z = 1 - 2 * v
    img = np.repeat((0.5 + 0.5 * z)[:, None, None], W, axis=1) * np.array([1.0, 2.0, 4.0])[None, None, :]
    e = o.env_create(img.astype(np.float32))
    rng = np.random.default_rng(12)
    for _ in range(200):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        if abs(d[2]) > 0.97:      # the equirect sampler WRAPS in v too (static sampler s1): the poles blend top and bottom rows
            continue
        df = d.astype(np.float32); out = np.zeros(3, np.float32)
        L.orc_sample_cube(o.h, e, df.ctypes.data_as(C.c_void_p), 0.0, out.ctypes.data_as(C.c_void_p))
        assert np.allclose(out, (0.5 + 0.5 * d[2]) * np.array([1, 2, 4]), atol=0.03)


# ---- BSDF ---------------------------------------------------------------------------------------------------
def surface(**kw):
    sp = dict(albedo=(0.8, 0.6, 0.4), alpha=1.0, metalness=0.0, roughness_squared=(0.25, 0.25), shading_normal=(0, 0, 1),
              anisotropy_tangent=(1, 0, 0), anisotropy_bitangent=(0, 1, 0), ior=1.5, specular_color=(1, 1, 1), specular_factor=1.0,
              clearcoat=0.0, clearcoat_roughness=0.1, clearcoat_normal=(0, 0, 1), sheen_color=(0, 0, 0), sheen_roughness_squared=0.25,
              transmissive=0.0, thickness=0.0, attenuation_distance=0.0, attenuation_color=(1, 1, 1))
    sp.update(kw)
    order = ["albedo", "alpha", "metalness", "roughness_squared", "shading_normal", "anisotropy_tangent", "anisotropy_bitangent", "ior",
             "specular_color", "specular_factor", "clearcoat", "clearcoat_roughness", "clearcoat_normal", "sheen_color",
             "sheen_roughness_squared", "transmissive", "thickness", "attenuation_distance", "attenuation_color"]
    flat = []
    for k in order:
        v = sp[k]
        flat += list(v) if hasattr(v, "__len__") else [v]
    assert len(flat) == 36
    return np.array(flat, np.float32)


def eval_bsdf(oracle_lib, o, flags, sp, v, l, ng=(0, 0, 1)):
    L = oracle_lib.lib()
    a = [np.array(x, np.float32) for x in (ng, v, l)]
    out = np.zeros(4, np.float32)
    L.orc_evaluate_bsdf(o.h, flags, sp.ctypes.data_as(C.c_void_p), a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p),
                        a[2].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out[:3], out[3]


def sample_bsdf(oracle_lib, o, flags, sp, u, v):
    L = oracle_lib.lib()
    uu, vv = np.array(u, np.float32), np.array(v, np.float32)
    out = np.zeros(9, np.float32)
    L.orc_sample_bsdf(o.h, flags, sp.ctypes.data_as(C.c_void_p), uu.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def test_diffuse_white_override(oracle_lib):
    o = oracle_lib.Oracle()
    sp = surface()
    v = np.array([0.3, 0.1, 0.9]); v /= np.linalg.norm(v)
    l = np.array([-0.2, 0.4, 0.7]); l /= np.linalg.norm(l)
    f, pdf = eval_bsdf(oracle_lib, o, abi.FLAG_MATERIAL_DIFFUSE_WHITE, sp, v, l)
    assert np.allclose(f, l[2] / math.pi, rtol=1e-6) and abs(pdf - l[2] / math.pi) < 1e-7
    r = sample_bsdf(oracle_lib, o, abi.FLAG_MATERIAL_DIFFUSE_WHITE, sp, (0.3, 0.6, 0.2), v)
    assert r[8] == 1 and r[7] == 0 and abs(r[0] / r[3] - 1.0) < 1e-5          # weight = bsdf/pdf = 1


def test_lambert_limit_and_energy(oracle_lib):
    """specular_factor 0, no metal: the layered BSDF reduces to albedo/pi * cos; its albedo integral is <= 1."""
    o = oracle_lib.Oracle()
    sp = surface(specular_factor=0.0)
    v = np.array([0, 0, 1.0])
    rng = np.random.default_rng(13)
    acc = np.zeros(3); n = 4000
    for _ in range(n):
        z = rng.random(); phi = rng.random() * 2 * math.pi; r = math.sqrt(1 - z * z)
        l = np.array([r * math.cos(phi), r * math.sin(phi), z])
        f, _ = eval_bsdf(oracle_lib, o, 0, sp, v, l)                           # no MIS flag: plain GltfBsdf * alpha
        assert np.allclose(f, np.array([0.8, 0.6, 0.4]) / math.pi * z, rtol=2e-4, atol=1e-7)
        acc += f * 2 * math.pi
    assert np.all(acc / n < 1.0)


def test_sample_pdf_consistency_mis(oracle_lib):
    """SampleBsdf with MATERIAL_MIS: the returned pdf equals EvaluateBsdf's pdf for the sampled direction, and
    E[1/pdf * indicator] recovers the hemisphere measure (pdf integrates to ~1 over the reflection lobes)."""
    o = oracle_lib.Oracle()
    sp = surface(roughness_squared=(0.3, 0.15), clearcoat=0.7, clearcoat_roughness=0.2, sheen_color=(0.5, 0.5, 0.5))
    v = np.array([0.4, -0.2, 0.89]); v /= np.linalg.norm(v)
    rng = np.random.default_rng(14)
    flags = abi.FLAG_MATERIAL_MIS
    for _ in range(300):
        r = sample_bsdf(oracle_lib, o, flags, sp, rng.random(3), v)
        l, pdf = r[4:7], r[3]
        if r[7] == 1 or not np.isfinite(pdf) or l[2] <= 0:
            continue
        f2, pdf2 = eval_bsdf(oracle_lib, o, flags, sp, v, l)
        assert abs(pdf - pdf2) <= 2e-4 * abs(pdf) + 1e-6
        assert np.allclose(r[:3], f2, rtol=2e-4, atol=1e-6)
    # numerical integral of the pdf over the upper hemisphere (all lobes reflect): close to 1, never above
    tot = 0.0; n = 20000
    for _ in range(n):
        z = rng.random(); phi = rng.random() * 2 * math.pi; rr = math.sqrt(1 - z * z)
        _, pdf = eval_bsdf(oracle_lib, o, flags, sp, v, np.array([rr * math.cos(phi), rr * math.sin(phi), z]))
        tot += pdf * 2 * math.pi
    assert 0.85 < tot / n < 1.05


def test_lobe_selection_order_and_alpha_lobe(oracle_lib):
    o = oracle_lib.Oracle()
    sp = surface(alpha=0.25)
    v = np.array([0, 0, 1.0])
    r = sample_bsdf(oracle_lib, o, abi.FLAG_MATERIAL_MIS, sp, (0.5, 0.1, 0.1), v)      # u.x <= 1 - alpha -> alpha lobe
    assert np.allclose(r[4:7], -v) and r[7] == 1 and r[8] == 0 and abs(r[3] - 0.75) < 1e-6 and np.allclose(r[:3], 0.75)
    r = sample_bsdf(oracle_lib, o, abi.FLAG_MATERIAL_MIS, sp, (0.9, 0.1, 0.1), v)      # past alpha + specular(0.125) -> diffuse
    assert r[7] == 0 and r[8] == 1 and r[6] > 0
    sp = surface(transmissive=1.0)
    r = sample_bsdf(oracle_lib, o, abi.FLAG_MATERIAL_MIS, sp, (0.9, 0.3, 0.4), v)      # specular 0.5, then transmission takes the rest
    assert r[7] == 1 and r[6] < 0                                                      # thin-walled: mirrored below the surface


def test_tonemap_reference_points(oracle_lib):
    L = oracle_lib.lib()
    cfg = abi.PtTonemapConfig(abi.TONEMAPPER_NONE, 1.0, 0, 0)
    for x in (0.0, 0.001, 0.0031308, 0.2, 0.5, 1.0, 3.0):
        rgb = np.array([x, x, x], np.float32); out = np.zeros(3, np.float32)
        L.orc_tonemap_pixel(C.byref(cfg), rgb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        c = min(max(x, 0.0), 1.0)
        expect = c * 12.92 if c <= 0.0031308 else 1.055 * c ** (1 / 2.4) - 0.055
        assert abs(out[0] - expect) < 2e-6
    cfg = abi.PtTonemapConfig(abi.TONEMAPPER_AGX, 1.0, 0, 0)
    prev = -1
    # AgX is monotonic on greys and stays in [0, 1].  (Below ~2.2e-4 the polynomial dips under 0 and pow(<0, 2.2) is NaN,
    # which the reference's UNORM render target turns into 0: the RGBA8 path saturates NaN to 0.)
    for x in (1e-3, 1e-2, 0.18, 1.0, 4.0, 16.0):
        rgb = np.array([x, x, x], np.float32); out = np.zeros(3, np.float32)
        L.orc_tonemap_pixel(C.byref(cfg), rgb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert 0 <= out[0] <= 1.0 and out[0] >= prev and abs(out[0] - out[1]) < 1e-3
        prev = out[0]


def test_defined_transcendentals_against_libm(oracle_lib):
    """atan2, log2, exp2, exp and pow are DEFINED by the oracle (hlsl.h o_atan2 ...: float kernels the HIP path states operation for operation,
    so that both sides produce the same bits); HLSL leaves their precision open.  Here: they are the functions they claim to be -- within a few
    ulp of libm's double results over wide ranges, with the special values of the C functions."""
    from ray_hook import math_inputs, oracle_math
    L = oracle_lib.lib()
    inp = math_inputs(n=200_000)

    def ulps(got, want):
        ok = np.isfinite(want) & (np.abs(want) > 1.2e-38) & (np.abs(want) < 3.4e38)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        return float((np.abs(got[ok].astype(np.float64) - want[ok]) / np.spacing(np.abs(want[ok].astype(np.float32))).astype(np.float64)).max())

    with np.errstate(all="ignore"):
        y, x = inp["atan2"]; g = oracle_math(L, "atan2", y, x); w = np.arctan2(y.astype(np.float64), x.astype(np.float64))
        assert ulps(g, w) < 2.5
        assert np.array_equal(np.signbit(g[~np.isnan(g)]), np.signbit(w[~np.isnan(w)]))
        a, _ = inp["log2"]; g = oracle_math(L, "log2", a, a); w = np.log2(a.astype(np.float64))
        far = np.abs(a - 1) > 0.01
        assert ulps(g[far], w[far]) < 3.5 and np.nanmax(np.abs(g[~far] - w[~far])) < 4e-9
        assert np.array_equal(np.isinf(g), np.isinf(w)) and np.array_equal(np.isnan(g), np.isnan(w))
        p, _ = inp["exp2"]; g = oracle_math(L, "exp2", p, p); w = 2.0 ** p.astype(np.float64)
        inside = (p >= -125) & (p < 127.99)
        assert ulps(g[inside], w[inside]) < 1.6 and np.all(g[p < -125] == 0) and np.all(np.isinf(g[p >= 128]))
        e, _ = inp["exp"]; g = oracle_math(L, "exp", e, e); w = np.exp(e.astype(np.float64))
        inside = (e >= -86.5) & (e <= 88.7)
        assert ulps(g[inside], w[inside]) < 2.0 and np.all(g[e < -86.5] == 0) and np.all(np.isinf(g[e > 88.75]))
        b, q = inp["pow"]; g = oracle_math(L, "pow", b, q); w = b.astype(np.float64) ** q.astype(np.float64)
        ok = np.isfinite(w) & (w > 1e-30) & (w < 1e30) & (b > 0) & np.isfinite(b) & np.isfinite(q)
        # pow = exp2(y * log2 x) in float: the error of log2 (3 ulp) times |y log2 x|, like the hardware instruction pair it stands for
        assert float((np.abs(g[ok] - w[ok]) / w[ok] / np.maximum(1.0, np.abs(q[ok] * np.log2(b[ok].astype(np.float64))))).max()) < 6e-7
        assert np.all(np.isnan(g[(b < 0)])) and np.all(g[(b == 0) & (q > 0)] == 0) and np.all(g[(b == 1) & np.isfinite(q)] == 1)
