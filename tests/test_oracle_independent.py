"""An independent check of the oracle's composite functions (CPU only; test infrastructure checking test infrastructure).

The oracle (oracle/shading.h, oracle/oracle.cpp) and the HIP kernels were written by the same hand from the same reading of the
HLSL, so a shared misreading would be invisible to the GPU parity tests.  Here the composite functions are restated a SECOND time,
directly from the reference's shader text, in numpy float64 and with a different structure (vectorised over the inputs, no shared
helpers with oracle/), and compared with liboracle's exports on thousands of random inputs:

  * GltfBsdf, both overloads                     Source/Shaders/Bsdf.hlsli:241-325 (and the helpers it calls, :26-228)
  * LayerProbabilities, BsdfPdf, EvaluateBsdf    Source/Shaders/PathTracer.lib.hlsl:535-590 (and the lobe pdfs :337-500)
  * SampleBsdf's pdf / value for its own direction  PathTracer.lib.hlsl:592-666
  * SampleImportanceMap, ImportanceMapPdf        Source/Shaders/Sampling.hlsli:123-174 (float32, operation for operation: exact)
  * GpuSkin's morph + skin arithmetic            Source/Shaders/Skin.cs.hlsl:61-128 (positions in float64; quirk q19)

float64 against the oracle's float32: a formula misread is an O(1) error, rounding is ~1e-6; the assertions sit between.
This reduces common-mode risk; it does not pin the oracle against real DXR output (still PARITY UNPINNED)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from gltf_renderer_amd import abi, scenes

PI = math.pi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUT = np.load(os.path.join(ROOT, "tests", "golden", "sheen_e_16x16.npy")).astype(np.float64).reshape(16, 16)      # rows = alpha, columns = cos(theta)


# ---- the second restatement (numpy, float64, arrays of N inputs) ---------------------------------------------------------------
def dot(a, b):
    return (a * b).sum(axis=-1)


def nrm(a):
    return a / np.sqrt(dot(a, a))[..., None]


def hclamp(x, lo, hi):                          # HLSL clamp = min(max(x, lo), hi), and min / max return the non-NaN operand
    return np.fmin(np.fmax(x, lo), hi)


def sat(x):                                      # saturate(NaN) = 0
    return hclamp(x, 0.0, 1.0)


def step(x):                                     # Heavyside, Bsdf.hlsli:29-32
    return (x > 0).astype(np.float64)


def hpow(x, y):                                  # HLSL pow = exp2(y * log2(x)): NaN for x < 0, pow(0, y > 0) = 0
    with np.errstate(all="ignore"):
        return np.exp2(y * np.log2(x))


def schlick(f0, c):                              # :39-47
    return f0 + (1 - f0) * hpow(1 - np.abs(c), 5.0)


def ggx_d(a, ndh):                               # :50-57
    a2 = a * a
    den = ndh * ndh * (a2 - 1) + 1
    return a2 * step(ndh) / (PI * den * den)


def ggx_corr_v(a, ndl, ndv, hdl, hdv):           # :77-84
    a2 = a * a
    den = np.abs(ndv) * np.sqrt(a2 + (1 - a2) * ndl * ndl) + np.abs(ndl) * np.sqrt(a2 + (1 - a2) * ndv * ndv)
    return 0.5 * step(hdl) * step(hdv) / den


def specular_brdf(a, ndl, ndv, ndh, hdl, hdv):   # :86-89
    return ggx_corr_v(a, ndl, ndv, hdl, hdv) * ggx_d(a, ndh)


def ggx_aniso_d(ax, ay, h):                      # :92-98
    a2 = ax * ay
    f = np.stack([ay * h[..., 0], ax * h[..., 1], a2 * h[..., 2]], axis=-1)
    w2 = a2 / dot(f, f)
    return step(h[..., 2]) * a2 * w2 * w2 / PI


def aniso_len(ax, ay, w):
    return np.sqrt((ax * w[..., 0]) ** 2 + (ay * w[..., 1]) ** 2 + w[..., 2] ** 2)


def aniso_specular(ax, ay, v, h, l):             # :116-129
    hdv, hdl = dot(h, v), dot(h, l)
    vv = np.abs(l[..., 2]) * aniso_len(ax, ay, v)
    ll = np.abs(v[..., 2]) * aniso_len(ax, ay, l)
    return 0.5 * step(hdv) * step(hdl) / (vv + ll) * ggx_aniso_d(ax, ay, h)


def fresnel_mix(f0_color, ior, weight, base, layer, hdv):            # :136-143
    f0 = ((1 - ior) / (1 + ior))[..., None] * np.ones(3)
    f0 = f0 * (f0 * f0_color)                    # `f0 *= f0 * f0_color`
    f0 = np.fmin(f0, 1.0)
    fr = schlick(f0, hdv[..., None])
    return (1 - weight[..., None] * fr.max(axis=-1)[..., None]) * base + weight[..., None] * fr * layer


def fresnel_coat(ior, weight, base, layer, ndv):                     # :156-162
    f0 = ((1 - ior) / (1 + ior)) ** 2
    fr = schlick(f0, ndv)
    t = (weight * fr)[..., None]
    return base + t * (layer - base)


def sheen_l(alpha, x):                           # :174-183
    t = (1 - alpha) ** 2
    lerp = lambda p, q: p + t * (q - p)
    a, b, c, d, e = lerp(21.5473, 25.3245), lerp(3.82987, 3.32435), lerp(0.19823, 0.16801), lerp(-1.97760, -1.27393), lerp(-4.32054, -4.85967)
    return a / (1 + b * hpow(x, c)) + d * x + e


def sheen_shadowing(alpha, c):                   # :185-192
    with np.errstate(all="ignore"):
        return np.where(c < 0.5, np.exp(sheen_l(alpha, c)), np.exp(2 * sheen_l(alpha, np.full_like(c, 0.5)) - sheen_l(alpha, 1 - c)))


def sheen_brdf(alpha, ndl, ndv, ndh):            # :164-202 (SheenBrdf hands (n_dot_v, n_dot_l) to SheenVisibility: symmetric)
    inv_r = 1 / alpha
    d = (2 + inv_r) * hpow(1 - ndh * ndh, inv_r * 0.5) / (2 * PI)
    with np.errstate(all="ignore"):
        vis = hclamp(1 / ((1 + sheen_shadowing(alpha, ndv) + sheen_shadowing(alpha, ndl)) * 4 * ndv * ndl), 0, 1)
    return d * vis


def sheen_e(alpha, c):
    """Texture2D<float>.SampleLevel(linear_clamp, (cos_theta, alpha), 0) on the 16x16 table: D3D bilinear, texel centres at +0.5."""
    x, y = c * 16 - 0.5, alpha * 16 - 0.5
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = x - x0, y - y0
    ix = lambda i: np.clip(i, 0, 15).astype(int)
    t = lambda i, j: LUT[ix(j), ix(i)]
    return (t(x0, y0) * (1 - fx) + t(x0 + 1, y0) * fx) * (1 - fy) + (t(x0, y0 + 1) * (1 - fx) + t(x0 + 1, y0 + 1) * fx) * fy


def modulate_roughness(a, ior):                  # :216-220
    return hclamp(a * sat(2 * (ior - 1)), 0.001, 1.0)


def thin_btdf(color, a, ior, n, v, l):           # :222-228
    a = modulate_roughness(a, ior)
    l = l - 2 * dot(n, l)[..., None] * n
    h = nrm(v + l)
    return color * specular_brdf(a, dot(n, l), dot(n, v), dot(n, h), dot(h, l), dot(h, v))[..., None]


def gltf_bsdf(sp, v, l, is_transmission=None):
    """Bsdf.hlsli:241-282 (is_transmission None) and :284-325 (a bool array)."""
    n = sp["shading_normal"]
    ax, ay = sp["roughness_squared"][..., 0], sp["roughness_squared"][..., 1]
    h = nrm(v + l)
    loc = lambda w: np.stack([dot(sp["anisotropy_tangent"], w), dot(sp["anisotropy_bitangent"], w), dot(n, w)], axis=-1)
    vl, hl, ll = loc(v), loc(h), loc(l)
    hdl, hdv = dot(h, l), dot(h, v)
    la = ll.copy(); la[..., 2] = np.abs(la[..., 2])
    h_dot_abs_l = dot(nrm(la + vl), vl)
    if is_transmission is None:
        refl = trans = np.ones(len(v))
    else:
        refl, trans = (~is_transmission).astype(np.float64), is_transmission.astype(np.float64)
    zeros = lambda x, keep: np.where(keep[..., None] > 0, x, 0.0)
    specular = zeros((sat(ll[..., 2]) * aniso_specular(ax, ay, vl, hl, ll))[..., None] * np.ones(3), refl)
    diffuse = zeros(sat(ll[..., 2])[..., None] * sp["albedo"] / PI, refl)
    transmission = zeros(sat(-ll[..., 2])[..., None] * thin_btdf(sp["albedo"], ay, sp["ior"], n, v, l), trans)
    diffuse = diffuse + sp["transmissive"][..., None] * (transmission - diffuse)
    dielectric = fresnel_mix(sp["specular_color"], sp["ior"], sp["specular_factor"], diffuse, specular, h_dot_abs_l)
    metal = zeros(specular * schlick(sp["albedo"], hdv[..., None]), refl)
    material = dielectric + sp["metalness"][..., None] * (metal - dielectric)
    sa = hclamp(sp["sheen_roughness_squared"], 0.000001, 1)
    sheen = zeros((sat(ll[..., 2]) * sheen_brdf(sa, ll[..., 2], vl[..., 2], hl[..., 2]))[..., None] * np.ones(3), refl)
    mx = sp["sheen_color"].max(axis=-1)
    scaling = np.fmin(1 - mx * sheen_e(sa, vl[..., 2]), 1 - mx * sheen_e(sa, ll[..., 2]))
    material = sp["sheen_color"] * sheen + material * scaling[..., None]
    cndv, cndh, cndl = dot(n, v), dot(n, h), dot(n, l)           # clearcoat lobe evaluated about the SHADING normal (:273-277)
    cc = np.where(refl > 0, sat(cndl) * specular_brdf(sp["clearcoat_roughness"], cndl, cndv, cndh, hdl, hdv), 0.0)
    return fresnel_coat(1.5, sp["clearcoat"], material, cc[..., None] * np.ones(3), cndv)


def layer_probabilities(sp, v):                  # PathTracer.lib.hlsl:535-553
    rem = np.ones(len(v))
    p = {}
    p["alpha"] = 1.0 - sp["alpha"]; rem = rem - p["alpha"]
    p["clearcoat"] = fresnel_coat(1.5, sp["clearcoat"], np.zeros((len(v), 3)), np.ones((len(v), 3)), dot(sp["clearcoat_normal"], v))[..., 0] * rem
    rem = rem - p["clearcoat"]
    p["sheen"] = np.where((sp["sheen_color"] > 0).any(axis=-1), 0.5, 0.0) * rem; rem = rem - p["sheen"]
    p["specular"] = 0.5 * rem; rem = rem - p["specular"]
    p["transmission"] = sp["transmissive"] * rem; rem = rem - p["transmission"]
    p["diffuse"] = rem
    return p


def bsdf_pdf(sp, v, l, is_transmission, p):      # :555-565 with the lobe pdfs :337-347, :355-358, :378-394, :401-404, :424-435
    n = sp["shading_normal"]
    ggx_normal_pdf = lambda a, nn, hh: ggx_d(a, dot(nn, hh)) * dot(nn, hh)                       # Sampling.hlsli:55-59
    # transmission: mirror l about the shading normal, isotropic GGX of the modulated roughness
    lt = l - 2 * dot(n, l)[..., None] * n
    ht = nrm(v + lt)
    with np.errstate(all="ignore"):
        t_pdf = ggx_normal_pdf(modulate_roughness(sp["roughness_squared"][..., 1], sp["ior"]), n, ht) / (4 * dot(v, ht))
        h = nrm(v + l)
        cc = ggx_normal_pdf(sp["clearcoat_roughness"], sp["clearcoat_normal"], h) / (4 * dot(v, h))
        cosine = sat(dot(l, n) / PI)
        hl = np.stack([dot(sp["anisotropy_tangent"], h), dot(sp["anisotropy_bitangent"], h), dot(n, h)], axis=-1)
        spec = ggx_aniso_d(sp["roughness_squared"][..., 0], sp["roughness_squared"][..., 1], hl) * hl[..., 2] / (4 * dot(v, h))
        refl = p["clearcoat"] * cc + p["sheen"] * cosine + p["specular"] * spec + p["diffuse"] * cosine
    return np.where(is_transmission, p["transmission"] * t_pdf, refl)


# ---- random inputs ------------------------------------------------------------------------------------------------------------------
ORDER = [("albedo", 3), ("alpha", 1), ("metalness", 1), ("roughness_squared", 2), ("shading_normal", 3), ("anisotropy_tangent", 3),
         ("anisotropy_bitangent", 3), ("ior", 1), ("specular_color", 3), ("specular_factor", 1), ("clearcoat", 1), ("clearcoat_roughness", 1),
         ("clearcoat_normal", 3), ("sheen_color", 3), ("sheen_roughness_squared", 1), ("transmissive", 1), ("thickness", 1),
         ("attenuation_distance", 1), ("attenuation_color", 3)]


def random_surfaces(rng, n):
    f32 = lambda x: np.asarray(x, np.float32).astype(np.float64)          # inputs representable in float32: both sides see the same numbers
    unit = lambda k: nrm(rng.normal(size=(k, 3)))
    sp = {}
    nn = unit(n)
    t = nrm(np.cross(nn, unit(n)))
    sp["shading_normal"], sp["anisotropy_tangent"], sp["anisotropy_bitangent"] = f32(nn), f32(t), f32(nrm(np.cross(t, nn)))
    sp["albedo"] = f32(rng.uniform(0.02, 1, (n, 3)))
    sp["alpha"] = f32(np.where(rng.random(n) < 0.5, 1.0, rng.uniform(0.1, 1, n)))
    sp["metalness"] = f32(np.where(rng.random(n) < 0.3, 0.0, rng.uniform(0, 1, n)))
    ry = np.maximum(rng.uniform(0.03, 1, n) ** 2, 0.001)
    sp["roughness_squared"] = f32(np.stack([np.maximum(ry + rng.uniform(0, 1, n) ** 2 * (1 - ry), 0.001), ry], axis=1))
    sp["ior"] = f32(rng.uniform(1.0, 2.0, n))
    sp["specular_color"] = f32(rng.uniform(0, 1, (n, 3)))
    sp["specular_factor"] = f32(rng.uniform(0, 1, n))
    sp["clearcoat"] = f32(np.where(rng.random(n) < 0.4, 0.0, rng.uniform(0, 1, n)))
    sp["clearcoat_roughness"] = f32(rng.uniform(0.03, 1, n))
    sp["clearcoat_normal"] = f32(nrm(nn + 0.2 * rng.normal(size=(n, 3))))
    sp["sheen_color"] = f32(np.where(rng.random((n, 1)) < 0.5, 0.0, rng.uniform(0, 1, (n, 3))))
    sp["sheen_roughness_squared"] = f32(np.maximum(rng.uniform(0.05, 1, n) ** 2, 0.001))
    sp["transmissive"] = f32(np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 1, n)))
    sp["thickness"], sp["attenuation_distance"] = np.zeros(n), np.zeros(n)
    sp["attenuation_color"] = np.ones((n, 3))
    return sp


def pack36(sp, i):
    flat = []
    for k, w in ORDER:
        flat += [float(sp[k][i])] if w == 1 else [float(x) for x in sp[k][i]]
    assert len(flat) == 36
    return np.array(flat, np.float32)


def directions(rng, sp, n, transmit_fraction):
    """v on the shading normal's side, l on the same side or (a fraction) through the surface; both float32-representable."""
    nn = sp["shading_normal"]
    def hemi(sign):
        d = nrm(rng.normal(size=(n, 3)))
        c = dot(d, nn)
        d = d - nn * (c - sign * np.abs(c))[..., None]              # reflect into the wanted hemisphere
        return nrm(d + sign * 0.05 * nn)                            # keep away from grazing, where Heavyside steps amplify rounding
    v = hemi(1.0)
    through = rng.random(n) < transmit_fraction
    l = np.where(through[:, None], hemi(-1.0), hemi(1.0))
    v32, l32 = v.astype(np.float32), l.astype(np.float32)
    v32 /= np.linalg.norm(v32, axis=1, keepdims=True); l32 /= np.linalg.norm(l32, axis=1, keepdims=True)
    return v32.astype(np.float64), l32.astype(np.float64), through


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want) / np.maximum(np.abs(want), 1e-4 * max(1.0, float(np.abs(want).max())) if want.size else 1.0)


N = 3000


@pytest.fixture(scope="module")
def orc(oracle_lib):
    o = oracle_lib.Oracle()
    yield oracle_lib.lib(), o
    o.close()


def _eval(L, o, flags, sp, v, l, ng):
    out = np.zeros((len(v), 4), np.float32)
    for i in range(len(v)):
        s36 = pack36(sp, i)
        a = [np.ascontiguousarray(x[i], np.float32) for x in (ng, v, l)]
        L.orc_evaluate_bsdf(o.h, flags, s36.ctypes.data_as(C.c_void_p), a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p),
                            a[2].ctypes.data_as(C.c_void_p), out[i].ctypes.data_as(C.c_void_p))
    return out[:, :3].astype(np.float64), out[:, 3].astype(np.float64)


def _check(name, got, want, tol_typical=2e-5, tol_max=2e-2, tol_median=1e-6):
    ok = np.isfinite(want) & np.isfinite(got)
    assert (np.isfinite(want) == np.isfinite(got)).mean() > 0.999, name
    e = rel_err(got[ok], want[ok])
    print("%s: median rel %.1e, 99%% %.1e, max %.1e over %d values" % (name, float(np.median(e)), float(np.quantile(e, 0.99)), float(e.max()), e.size))
    assert np.median(e) <= tol_median, (name, float(np.median(e)))
    assert np.quantile(e, 0.99) <= tol_typical, (name, float(np.quantile(e, 0.99)))
    assert e.max() <= tol_max, (name, float(e.max()))


def test_gltf_bsdf_without_the_transmission_flag(orc):
    """EvaluateBsdf with FLAG_MATERIAL_MIS off = alpha * GltfBsdf(sp, v, l) (Bsdf.hlsli:241-282), pdf = alpha * saturate(n.l) / pi."""
    L, o = orc
    rng = np.random.default_rng(101)
    sp = random_surfaces(rng, N)
    v, l, _ = directions(rng, sp, N, 0.35)
    got, pdf = _eval(L, o, 0, sp, v, l, sp["shading_normal"])
    want = sp["alpha"][:, None] * gltf_bsdf(sp, v, l)
    _check("GltfBsdf (all lobes)", got, want)
    _check("cosine pdf", pdf, sat(dot(sp["shading_normal"], l)) / PI * sp["alpha"], 2e-6)
    assert (want.max(axis=1) > 1e-3).mean() > 0.6                   # the comparison is not vacuous


def test_gltf_bsdf_with_the_transmission_flag_layer_probabilities_and_pdf(orc):
    """EvaluateBsdf with FLAG_MATERIAL_MIS: is_transmission from the geometric normal, LayerProbabilities, BsdfPdf, and
    alpha * GltfBsdf(sp, v, l, is_transmission) (PathTracer.lib.hlsl:567-590, Bsdf.hlsli:284-325)."""
    L, o = orc
    rng = np.random.default_rng(102)
    sp = random_surfaces(rng, N)
    v, l, through = directions(rng, sp, N, 0.4)
    ng = sp["shading_normal"]                                        # geometric = shading normal here: `through` IS is_transmission
    got, pdf = _eval(L, o, abi.FLAG_MATERIAL_MIS, sp, v, l, ng)
    is_t = dot(ng, l) * dot(ng, v) < 0
    assert np.array_equal(is_t, through)
    want = sp["alpha"][:, None] * gltf_bsdf(sp, v, l, is_t)
    _check("GltfBsdf (is_transmission overload)", got, want)
    p = layer_probabilities(sp, v)
    assert np.allclose(sum(p.values()), 1.0, atol=1e-12)
    _check("BsdfPdf", pdf, bsdf_pdf(sp, v, l, is_t, p))
    assert (is_t & (want.max(axis=1) > 1e-4)).sum() > 100 and (~is_t & (want.max(axis=1) > 1e-3)).sum() > 1000


def test_sample_bsdf_returns_its_own_pdf_and_value_and_picks_the_lobe_by_the_probabilities(orc):
    """SampleBsdf (PathTracer.lib.hlsl:592-666): the lobe is chosen by walking u.x through alpha / clearcoat / sheen / specular /
    transmission / diffuse in that order; value and pdf are GltfBsdf / BsdfPdf of the sampled direction."""
    L, o = orc
    rng = np.random.default_rng(103)
    n = 2000
    sp = random_surfaces(rng, n)
    v, _, _ = directions(rng, sp, n, 0.0)
    u = rng.random((n, 3)).astype(np.float32)
    out = np.zeros((n, 9), np.float32)
    for i in range(n):
        s36 = pack36(sp, i); vv = np.ascontiguousarray(v[i], np.float32)
        L.orc_sample_bsdf(o.h, abi.FLAG_MATERIAL_MIS, s36.ctypes.data_as(C.c_void_p), u[i].ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p),
                          out[i].ctypes.data_as(C.c_void_p))
    p = layer_probabilities(sp, v)
    ux = u[:, 0].astype(np.float64)
    edges = np.cumsum(np.stack([p["alpha"], p["clearcoat"], p["sheen"], p["specular"], p["transmission"]], axis=1), axis=1)
    margin = np.abs(edges - ux[:, None]).min(axis=1) > 1e-5         # away from a lobe boundary (float32 subtraction chain upstream)
    alpha_lobe = ux <= edges[:, 0]
    trans_lobe = (ux > edges[:, 3]) & (ux <= edges[:, 4])
    is_t = out[:, 7] == 1
    assert np.array_equal(is_t[margin], (alpha_lobe | trans_lobe)[margin])
    assert np.array_equal((out[:, 8] == 0)[margin], alpha_lobe[margin])                          # use_mis off only for the alpha lobe
    a = margin & alpha_lobe
    assert a.sum() > 100
    assert np.allclose(out[a, 4:7], -v[a], atol=1e-6) and np.allclose(out[a, 3], p["alpha"][a], atol=1e-6)
    assert np.allclose(out[a, 0], 1 - sp["alpha"][a], atol=1e-6)
    # a sampled direction sits ON its lobe's peak, where D(h) of a near-mirror lobe amplifies the float32 rounding of h = normalize(v + l) by
    # ~1 / alpha^2: compared where every lobe's alpha is >= 0.05 (the evaluation of sharper lobes is covered off-peak by the tests above)
    smooth = np.minimum(sp["roughness_squared"].min(axis=1), np.minimum(sp["clearcoat_roughness"], modulate_roughness(sp["roughness_squared"][:, 1], sp["ior"]))) >= 0.05
    b = margin & ~alpha_lobe & np.isfinite(out).all(axis=1) & smooth
    assert b.sum() > 500
    l = out[b, 4:7].astype(np.float64)
    sub = {k: x[b] for k, x in sp.items()}
    pb = {k: x[b] for k, x in p.items()}
    _check("SampleBsdf value", out[b, :3], sub["alpha"][:, None] * gltf_bsdf(sub, v[b], l, is_t[b]), 1e-4)
    _check("SampleBsdf pdf", out[b, 3], bsdf_pdf(sub, v[b], l, is_t[b], pb), 1e-4)


def test_importance_map_sampling_and_pdf_operation_for_operation(oracle_lib):
    """SampleImportanceMap / ImportanceMapPdf (Sampling.hlsli:123-174) in float32, one operation at a time: bit-exact."""
    L = oracle_lib.lib()
    o = oracle_lib.Oracle()
    rng = np.random.default_rng(104)
    H, W = 128, 256
    img = (rng.random((H, W, 3)) ** 6 * 50 + 0.01).astype(np.float32)
    img[40:44, 100:104] = 5000.0
    e = o.env_create(img)
    n, cube, pyr = o.env_read(e)
    levels, off, r = [], 0, 1024
    while r >= 1:
        levels.append(pyr[off:off + r * r].reshape(r, r)); off += r * r; r //= 2
    mips = len(levels)
    assert mips == 11
    f = np.float32
    us = rng.random((1500, 2)).astype(np.float32)
    us[:20, 0] = 0.0; us[20:40, 1] = np.nextafter(f(1), f(0))
    mismatches = 0
    for u in us:
        ux, uy = f(u[0]), f(u[1])
        px = py = 0
        for i in range(mips - 2, -1, -1):
            px <<= 1; py <<= 1
            m = levels[i]
            ul, ur, ll, lr = m[py, px], m[py, px + 1], m[py + 1, px], m[py + 1, px + 1]
            left, right = f(ul + ll), f(ur + lr)
            total = f(left + right)
            prob_left = f(left / total)
            if ux < prob_left:
                ux = f(ux / prob_left)
                pu = f(ul / left)
            else:
                px += 1
                ux = f(f(ux - prob_left) / f(f(1) - prob_left))
                pu = f(ur / right)
            if uy < pu:
                uy = f(uy / pu)
            else:
                py += 1
                uy = f(f(uy - pu) / f(f(1) - pu))
        pdf = f(f(f(f(1024) * f(1024)) * levels[0][py, px]) / levels[mips - 1][0, 0])
        uv = np.array([f(f(f(px) + ux) / f(1024)), f(f(f(py) + uy) / f(1024))], np.float32)
        got = np.zeros(3, np.float32)
        L.orc_sample_importance_map(o.h, e, np.ascontiguousarray(u).ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p))
        if not (np.array_equal(got[:2], uv) and got[2] == pdf):
            mismatches += 1
        # ImportanceMapPdf at that uv: UVToPixel's off-by-one texel (quirk q9), both axes scaled by the WIDTH
        qx, qy = int(np.floor(f(uv[0] * f(1024))) - 0.5), int(np.floor(f(uv[1] * f(1024))) - 0.5)
        val = levels[0][qy, qx] if 0 <= qx < 1024 and 0 <= qy < 1024 else f(0)
        want = f(f(f(f(1024) * f(1024)) * val) / levels[mips - 1][0, 0])
        L.orc_importance_map_pdf.restype = C.c_float
        assert L.orc_importance_map_pdf(o.h, e, uv.ctypes.data_as(C.c_void_p)) == want
    assert mismatches == 0, mismatches
    o.close()


# ---- GpuSkin: morph targets (Skin.cs.hlsl:61-128) ------------------------------------------------------------------------------------
def _oracle_skin(oracle_lib, s, morph, bones, in_flags=None, t=0.37):
    o = oracle_lib.Oracle()
    h = s.upload(o)
    bind = scenes.SkinBinding(o, s, h, 0, 0, morph=morph)
    if in_flags is not None:
        bind.params.input_mesh_flags = in_flags
    bind.pose(t, bones=bones)
    nv = s.skins[0]["mesh"].num_vertices
    out = (o.buffer_read(bind.out_position, np.float32, nv * 3).reshape(-1, 3), o.buffer_read(bind.out_tangent_space, np.uint32, nv))
    o.close()
    return out


def _decode_ts(oracle_lib, packed):
    L = oracle_lib.lib()
    n, t = np.zeros((len(packed), 3), np.float32), np.zeros((len(packed), 4), np.float32)
    for i, p in enumerate(packed):
        L.orc_decode_tangent_space(C.c_uint32(int(p)), n[i].ctypes.data_as(C.c_void_p), t[i].ctypes.data_as(C.c_void_p))
    return n.astype(np.float64), t.astype(np.float64)


@pytest.mark.parametrize("bones", [True, False])
def test_oracle_morph_targets_against_a_float64_restatement(oracle_lib, bones):
    """Positions: p' = sum_i w_i B_i (p + sum_k m_k dP_k).  Normals: with bones, the skinned normal of (n + sum m_k dN_k); without bones
    the input flags are all cleared (quirk q19), so the output normal is normalize(sum m_k dN_k) alone."""
    s = scenes.skinned_figure(32, 18)
    scenes.add_morph_targets(s)
    sk = s.skins[0]; mesh = sk["mesh"]
    morph = [(0, 0.4), (1, 0.15), (4, 0.9), (2, 0.6)]
    pos, ts = _oracle_skin(oracle_lib, s, morph, bones)
    p = mesh.positions.astype(np.float64).copy()
    for ti, w in morph:
        src = sk["targets"][ti]["sources"]
        if "POSITION" in src:
            p += float(np.float32(w)) * src["POSITION"].astype(np.float64)
    # morph normals as the shader sees them: the packed 10-10-10-2 target streams, decoded
    dn = np.zeros_like(p)
    for ti, w in morph:
        b = sk["targets"][ti]["tangent_space"]
        if b != -1:
            dn += float(np.float32(w)) * _decode_ts(oracle_lib, s.buffers[b][0])[0]
    if bones:
        B = scenes.bones_for_pose(sk, np.eye(4), scenes.skinned_figure_pose(0.37))
        M = np.stack([np.array(b.transform[:], np.float64).reshape(4, 4).T for b in B])
        IT = np.stack([np.array(b.inverse_transpose[:], np.float64).reshape(4, 4).T for b in B])
        jw = scenes.meshgen.pack_joint_weight(mesh.joints, mesh.weights)
        ids = jw[:, :4].astype(int); w = jw[:, 4:].astype(np.float64) / 65535.0
        ph = np.concatenate([p, np.ones((len(p), 1))], axis=1)
        n0 = _decode_ts(oracle_lib, mesh.tangent_space_stream())[0] + dn
        want_p, want_n = np.zeros_like(p), np.zeros_like(p)
        for k in range(4):
            want_p += w[:, k:k + 1] * np.einsum("nij,nj->ni", M[ids[:, k]], ph)[:, :3]
            want_n += w[:, k:k + 1] * np.einsum("nij,nj->ni", IT[ids[:, k]][:, :3, :3], n0)
    else:
        want_p, want_n = p, dn
    assert np.abs(pos - want_p).max() < 5e-6
    got_n = _decode_ts(oracle_lib, ts)[0]
    cosang = dot(got_n, nrm(want_n))
    assert np.quantile(cosang, 0.01) > 0.9995 and cosang.min() > 0.99, (float(cosang.min()), float(np.quantile(cosang, 0.01)))      # 10-bit octahedral quantisation
    # and the targets moved things: not the rest mesh
    assert np.abs(pos - mesh.positions).max() > 0.02


def test_pick_morph_targets_rule():
    """Renderer.cpp:425-443 restated in scenes.pick_morph_targets (host logic used by the GPU morph test through gs_frame)."""
    assert scenes.pick_morph_targets([0.0, -1.0, 0.2]) == [(2, pytest.approx(0.2))]
    assert [t for t, _ in scenes.pick_morph_targets([0.1, 0.2, 0.3, 0.4, 0.05])] == [0, 1, 2, 3]          # smaller than every held weight
    assert [t for t, _ in scenes.pick_morph_targets([0.1, 0.2, 0.3, 0.4, 0.5])] == [4, 1, 2, 3]           # replaces the smallest
    assert [t for t, _ in scenes.pick_morph_targets([0.3, 0.1, 0.1, 0.4, 0.2, 0.25])] == [0, 4, 5, 3]     # first of equal minima first


def test_multi_threaded_cpu_lbvh_is_the_single_threaded_tree(oracle_lib):
    """bench.py's cpu_baseline times the CPU LBVH build on one core and on all of them (leg B2): the parallel build
    (chunk sorts + stable merges, subtrees built by different threads into one pre-order-numbered array) must be the SAME tree -- same node
    and triangle-test counts for the same rays, same image bit for bit."""
    s = scenes.sponza_class(width=64, height=36, tex=16)
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
    res = []
    for threads in (1, 8, 3):
        o = oracle_lib.Oracle(); h = s.upload(o)
        o.build_accel(threads)
        img = np.zeros((s.height, s.width, 4), np.float32)
        o.counters()
        o.trace(st, s.execute_params(0, env_handle=h["env"]), img, nthreads=4)
        c = o.counters()
        res.append((img, c["nodes"], c["tris"], c["rays"], o.bvh_info()))
        o.close()
    for img, nodes, tris, rays, info in res[1:]:
        assert np.array_equal(img, res[0][0]) and (nodes, tris, rays, info) == res[0][1:]
    assert res[0][4][1] > 200000                                     # the bench scene's triangle count, not a toy


@pytest.mark.parametrize("scene", ["sponza", "grid"])
def test_oracle_tree_finds_what_the_exhaustive_search_finds(oracle_lib, scene):
    """The oracle's LBVH is only an accelerator: for every ray it has to return the hit that testing EVERY triangle returns -- same triangle,
    same t, u, v to the bit.  That holds because of two rules of its triangle test (oracle.cpp Tracer::intersect): a candidate must pass the
    box test of its own box (else a tree could cull a ray that the float test would still accept a few ulp outside the triangle), and of two
    triangles at exactly the same distance the lower (instance, primitive) wins.  Rays as the path tracer casts them: from points ON surfaces,
    in random, axis-aligned and grazing directions, with and without face culling."""
    from ray_hook import surface_rays
    s = scenes.sponza_class(width=64, height=36, tex=16) if scene == "sponza" else scenes.material_grid(64, 36)
    o = oracle_lib.Oracle(); s.upload(o)
    first, second = surface_rays(o, s, 60_000, 7)
    rays = np.concatenate([first[:1500], second[:6000]])
    for flags in (0, 0x10, 0x20):
        o.set_brute_force(False); tree = o.intersect_many(rays, flags, 0)
        o.set_brute_force(True); every = o.intersect_many(rays, flags, 0)
        assert np.array_equal(tree.view(np.uint32), every.view(np.uint32)), int((tree.view(np.uint32) != every.view(np.uint32)).any(axis=1).sum())
        assert 0.2 < tree[:, 0].mean() < 0.95
    o.close()
