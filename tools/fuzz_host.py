"""Builds the sanitizer-instrumented fuzzer of the plain-C++ scene side (tests/fuzz/host_fuzz.cpp), writes seed files and runs it.
usage: python tools/fuzz_host.py [iterations] [rng seed] [seed-name filter, e.g. sink]        (CPU only; test infrastructure)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(exe):
    src = [os.path.join(ROOT, p) for p in ("tests/fuzz/host_fuzz.cpp", "gltf_renderer_amd/csrc/host/image_decode.cpp", "gltf_renderer_amd/csrc/host/gltf_scene.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "include")] + src + ["-o", exe])


def write_seeds(d):
    """The kitchen-sink scene in two containers (embedded: the fuzzer mutates one file) and one image for every decoder path."""
    import numpy as np
    import test_gltf_loader as tl
    tl.build_kitchen_sink("view").write_glb(os.path.join(d, "sink.glb"))
    tl.build_kitchen_sink("uri").write_gltf(os.path.join(d, "sink.gltf"))
    rng = np.random.default_rng(5)
    rgba, rgb = tl.smooth_image(rng, 37, 53, 4), tl.smooth_image(rng, 41, 29, 3)
    def put(name, data):
        open(os.path.join(d, name), "wb").write(data)
    put("rgba.png", tl.png_bytes(rgba)); put("grey.png", tl.png_bytes(rgb[..., 0]))
    put("base.jpg", tl.jpeg_bytes(rgb, quality=85)); put("prog.jpg", tl.jpeg_bytes(rgb, quality=70, progressive=True, subsampling=2))
    put("grey.jpg", tl.jpeg_bytes(rgb[..., 0]))
    f = (rng.random((19, 31, 3)) * 4).astype(np.float32)
    put("rle.hdr", tl.hdr_file(tl.rgbe_encode(f), True)); put("flat.hdr", tl.hdr_file(tl.rgbe_encode(f), False))
    for comp in (0, 1, 2, 3):
        put("c%d_half.exr" % comp, tl.exr_file(f, np.float16, comp)); put("c%d_float.exr" % comp, tl.exr_file(f, np.float32, comp, with_alpha=True))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        exe = os.path.join(d, ".host_fuzz")
        build(exe)
        write_seeds(d)
        print("seeds:", sorted(n for n in os.listdir(d) if not n.startswith(".")))
        r = subprocess.run([exe, d, str(iters), str(seed)] + sys.argv[3:4], capture_output=True, text=True)
        sys.stdout.write(r.stdout[-2000:]); sys.stderr.write(r.stderr[-6000:])
        return r.returncode


if __name__ == "__main__":
    sys.exit(main())
