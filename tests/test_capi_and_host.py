"""CPU tests of the boundary (no GPU): the C-ABI library builds/loads and exports
every symbol include/pcm_amd.h declares, struct layouts match the header, the
host-side float math of the plane fit equals the oracle bit for bit, the product
path fails loudly without a GPU, and the product never touches oracle/."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pcm):
    pcm.build_library()
    L = pcm.load_library()
    hdr = open(os.path.join(ROOT, "include", "pcm_amd.h")).read()
    declared = set(re.findall(r"\b(pcm_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"pcm_ctx"}
    assert len(declared) >= 20
    from pointcloud_slam_amd import capi
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pcm_abi_version() == 3


def test_struct_layouts_match_header(pcm):
    from pointcloud_slam_amd import capi
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "pcm_amd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(pcm_config), sizeof(pcm_result), sizeof(pcm_stats),
             offsetof(pcm_config, voxel_resolution), offsetof(pcm_config, flags), offsetof(pcm_result, H), offsetof(pcm_result, status));
      return 0;
    }'''
    exe = "/tmp/pcm_layout_check"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(capi.PcmConfig), C.sizeof(capi.PcmResult), C.sizeof(capi.PcmStats), capi.PcmConfig.voxel_resolution.offset,
            capi.PcmConfig.flags.offset, capi.PcmResult.H.offset, capi.PcmResult.status.offset]
    assert got == want


def test_flag_constants_match_header(pcm):
    """Every PCM_FLAG_* / PCM_ERR_* / PCM_COV_* constant of the Python binding has the value include/pcm_amd.h defines."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pcm_amd.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(PCM_(?:FLAG|ERR|COV)_[A-Z0-9_]+)\s+(-?\d+)", hdr)}
    assert {"PCM_FLAG_NEIGHBOUR_LISTS", "PCM_FLAG_NO_NEIGHBOUR_LISTS", "PCM_FLAG_REFERENCE_KNN_ORDER", "PCM_FLAG_LIO_REFERENCE_SEMANTICS"} <= set(defs)
    seen = 0
    for name, value in defs.items():
        if hasattr(pcm.capi, name):
            assert getattr(pcm.capi, name) == value, name
            seen += 1
    assert seen >= 7
    flags = [v for k, v in defs.items() if k.startswith("PCM_FLAG_")]
    assert len(set(flags)) == len(flags) and all(v & (v - 1) == 0 for v in flags)   # distinct single bits


def test_default_config_is_the_reference_defaults(pcm):
    from pointcloud_slam_amd import capi
    L = pcm.load_library()
    cfg = capi.PcmConfig()
    L.pcm_default_config(C.byref(cfg))
    assert (cfg.optimizer, cfg.max_iterations, cfg.lm_max_iterations) == (1, 64, 10)      # lsq_registration_impl.hpp:11-17
    assert (cfg.rotation_eps, cfg.translation_eps, cfg.lm_init_lambda_factor) == (2e-3, 5e-4, 1e-9)
    assert (cfg.knn, cfg.min_knn) == (5, 3) and abs(cfg.plane_threshold - 0.1) < 1e-7          # options.h:14-15, options.cc:10
    assert cfg.max_range == 5.0 and cfg.k_correspondences == 20 and cfg.regularization == 3  # ivox3d.h:80, fast_gicp_impl.hpp:16,20


def test_no_gpu_fails_loudly(pcm):
    """There is no CPU fallback: without a HIP device every compute entry point errors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pcm.PcmError) as e:
        pcm.P2PlaneRegistration(0)
    assert "no CPU fallback" in str(e.value) or e.value.code == -3


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pointcloud-slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_" not in text and "libpcm_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
    for f in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", f)
        if os.path.isfile(p):
            assert "oracle" not in open(p).read()


def test_plane_fit_host_math_equals_oracle_bitwise():
    """csrc/plane_fit.h (the kernel's plane fit, compiled here for the host with g++)
    against the oracle's restatement on 60k random neighbourhoods, m = 3, 4, 5."""
    from oracle import build
    build()
    src = r'''
    #include "plane_fit.h"
    #include <cstdio>
    #include <cstdlib>
    extern "C" int orc_test_esti_plane(const float *pts_xyz, int n, float threshold, float plane[4]);
    int main() {
      srand(7);
      int nbad = 0;
      for (int t = 0; t < 60000; t++) {
        float px[5], py[5], pz[5], pts[15];
        float nx = rand() / (float)RAND_MAX - 0.5f, ny = rand() / (float)RAND_MAX - 0.5f, nz = rand() / (float)RAND_MAX - 0.5f;
        float nn = sqrtf(nx * nx + ny * ny + nz * nz); nx /= nn; ny /= nn; nz /= nn;
        float ox = 100.f * (rand() / (float)RAND_MAX - 0.5f), oy = 100.f * (rand() / (float)RAND_MAX - 0.5f), oz = 10.f * (rand() / (float)RAND_MAX - 0.5f);
        for (int j = 0; j < 5; j++) {
          float a = rand() / (float)RAND_MAX - 0.5f, b = rand() / (float)RAND_MAX - 0.5f, c = rand() / (float)RAND_MAX - 0.5f;
          float d = a * nx + b * ny + c * nz;
          px[j] = ox + a - d * nx + 0.02f * (rand() / (float)RAND_MAX - 0.5f); py[j] = oy + b - d * ny; pz[j] = oz + c - d * nz;
          pts[3 * j] = px[j]; pts[3 * j + 1] = py[j]; pts[3 * j + 2] = pz[j];
        }
        int m = 3 + t % 3;
        float4 pl; float po[4];
        bool ok = pcm::esti_plane(px, py, pz, m, 0.1f, &pl);
        int ok2 = orc_test_esti_plane(pts, m, 0.1f, po);
        if (ok != (bool)ok2 || pl.x != po[0] || pl.y != po[1] || pl.z != po[2] || pl.w != po[3]) nbad++;
      }
      printf("%d\n", nbad);
      return 0;
    }'''
    exe = "/tmp/pcm_planefit_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-x", "c++", "-", "-I", os.path.join(ROOT, "pointcloud-slam_amd", "csrc"),
                    "-L", os.path.join(ROOT, "oracle"), "-lpcm_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", exe],
                   input=src.encode(), check=True)
    assert int(subprocess.check_output([exe]).strip()) == 0


def test_lsq_step_host_matches_oracle(synth):
    """csrc/lsq_step.h (the GN / LM state machine the device runs: left-looking pivoted LDL^T, so3_exp, convergence test,
    LM rho / lambda / nu bookkeeping) is compiled for the HOST with g++ and fed the oracle's trace of an align() -- every
    linearize (cost, H, b) and every trial cost in order: the replay must end in the oracle's pose bit for bit, with the same
    iteration, linearize and compute_error counts and the same convergence flag, for GN and for LM."""
    from oracle import Oracle, build
    from oracle.loader import result_T
    build()
    src = r'''
    #include "lsq_step.h"
    #include <cstdio>
    #include <cstdlib>
    using namespace pcm;
    int main(int argc, char** argv) {
      // stdin: optimizer max_iterations lm_max_iterations rot_eps trans_eps lm_init_lambda ; 16 guess floats ; n ; n x 43 records
      LsqParams lp; double tmp;
      if (scanf("%d %d %d %lf %lf %lf", &lp.optimizer, &lp.max_iterations, &lp.lm_max_iterations, &lp.rotation_eps, &lp.translation_eps, &lp.lm_init_lambda_factor) != 6) return 2;
      float guess[16];
      for (int i = 0; i < 16; i++) { if (scanf("%lf", &tmp) != 1) return 2; guess[i] = (float)tmp; }
      int n; if (scanf("%d", &n) != 1) return 2;
      PairState s; init_state(s, guess);
      int used = 0;
      for (int r = 0; r < n && s.mode != MODE_DONE; r++) {
        double rec[43];
        for (int k = 0; k < 43; k++) { char tok[64]; if (scanf("%63s", tok) != 1) return 2; rec[k] = strtod(tok, nullptr); }
        const bool trial = rec[1] != rec[1];
        if (trial != (s.mode == MODE_TRIAL)) { printf("ORDER %d\n", r); return 1; }
        if (trial) after_trial(s, lp, rec[0]);
        else after_linearize(s, lp, rec + 1, rec + 37, rec[0], 0);
        used++;
      }
      printf("%d %d %d %d %d %d\n", used, s.mode == MODE_DONE, s.iter, s.converged, s.num_linearize, s.num_compute_error);
      for (int i = 0; i < 16; i++) printf("%a\n", s.x0[i]);
      printf("%a %a\n", s.lambda, s.nu);
      return 0;
    }'''
    exe = "/tmp/pcm_lsq_step_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-x", "c++", "-", "-I", os.path.join(ROOT, "pointcloud-slam_amd", "csrc"), "-o", exe],
                   input=src.encode(), check=True)
    p = synth.make_pair(5, 3000, 40000)
    for opt, oid in (("GN", 0), ("LM", 1)):
        for factor in ((1e-9,) if opt == "GN" else (1e-9, 1e-2, 10.0)):   # large initial lambdas force rejected LM trials (rho < 0 branches)
            o = Oracle("P2PLANE", opt, voxel_resolution=0.5, num_neighbors=27, lm_init_lambda_factor=factor)
            o.set_input_target(p.submap); o.set_input_source(p.scan)
            o.enable_trace(512)
            r = o.align(p.guess)
            tr = o.trace()
            assert len(tr) == r.num_linearize + r.num_compute_error
            text = "%d %d %d %r %r %r\n" % (oid, 64, 10, 2e-3, 5e-4, factor)
            text += " ".join(repr(float(v)) for v in np.asarray(p.guess, np.float32).reshape(-1)) + "\n%d\n" % len(tr)
            text += "\n".join(" ".join(float(v).hex() if v == v else "nan" for v in row) for row in tr) + "\n"
            out = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split()
            used, done, it, conv, nlin, nce = (int(v) for v in out[:6])
            x = np.array([float.fromhex(v) for v in out[6:22]]).reshape(4, 4)
            assert used == len(tr) and done == 1
            assert it == r.iterations and conv == int(r.converged) and nlin == r.num_linearize and nce == r.num_compute_error
            assert np.array_equal(x, result_T(r)), (opt, factor, np.abs(x - result_T(r)).max())
        if opt == "LM":
            assert r.num_compute_error >= r.num_linearize
