"""Static checks on the gfx950 ISA hipcc emits for the library (no GPU needed): the source is compiled exactly as
__graft_entry__.build() compiles it and the device assembly is inspected.

1. NO packed fp32 VALU instruction with a crossed operand select anywhere in the library.  On MI355X a
   v_pk_mul/fma/add_f32 whose low result reads the HIGH half of a source (op_sel:[..,1,..]) returns a wrong low result
   in lanes 48..63 when another wave of the SIMD issues bf16 MFMAs (scripts/ubench/pk_mfma.hip, exact integer
   arithmetic; profiles/r08b_packed_fp32_beside_mfma.md).  hipcc's SLP vectoriser emits exactly that form for
   "broadcast weight x pair of texels"; the library is therefore built with -fno-slp-vectorize (csrc/lrf_tu.h).  This
   was the cause of the run-to-run differences of rounds 1-2 (docs/GFX950_FINDINGS.md finding 17).
2. k_shade3 (the default colour kernel, and with SAVE the row-saving forward of the training step): 135
   v_mfma_f32_32x32x16_bf16 per tile (15 basis + 24 layer 1 + 96 layer 2), no scratch, at most 256 registers (two waves per SIMD).
3. k_march and the two 32-sample kernels of the colour
   network's backward (k_train_dgrad3, k_train_app3) use no scratch and at most 256 registers; the weight-gradient kernel
   keeps its shape.
"""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "localrf_amd", "csrc", "lrf_render.hip")
BUILD_FLAGS = ["-fno-slp-vectorize"]            # = __graft_entry__.build()


def _device_asm(extra):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17"] + extra + ["-I",
                               os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, SRC],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


@pytest.fixture(scope="module")
def asm():
    import __graft_entry__ as ge
    src = open(ge.__file__).read()
    assert all(f in src for f in BUILD_FLAGS), "tests/test_isa_checks.py must compile with the flags of __graft_entry__.build()"
    return _device_asm(BUILD_FLAGS)


def _kernels(asm):
    """name -> body text of every kernel in the assembly."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        out[m[1]] = m[2]
    return out


def _body(asm, pattern):
    hits = [(k, v) for k, v in _kernels(asm).items() if re.search(pattern, k)]
    assert hits, pattern
    return hits


def test_no_packed_fp32_with_crossed_operand_select(asm):
    """The erratum-prone form: v_pk_{mul,fma,add}_f32 with any op_sel bit set (a LOW result reading a HIGH source half).
    op_sel_hi deviations (a high result reading a low half) were never wrong in the exact test and are allowed."""
    n_kernels = n_pk = 0
    for name, body in _kernels(asm).items():
        n_kernels += 1
        for ln in body.splitlines():
            if not re.search(r"\bv_pk_(mul|fma|add)_f32\b", ln):
                continue
            n_pk += 1
            m = re.search(r"op_sel:\[([01,]+)\]", ln)
            assert not (m and "1" in m[1]), (name, ln.strip())
    assert n_kernels > 40, n_kernels
    print("kernels", n_kernels, "packed fp32 instructions (all with straight low halves)", n_pk)


def test_build_does_not_use_the_slp_vectoriser():
    import __graft_entry__ as ge
    src = open(ge.__file__).read()
    assert "-fno-slp-vectorize" in src


def test_shade3_shape(asm):
    hits = _body(asm, r"k_shade3ILi8ELb[01]ELb0ELb[01]EE")                     # LDS / global tile offsets x eval / SAVE
    assert len(hits) == 4, [k for k, _ in hits]
    for name, body in hits:
        assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)) == 135, name
        assert not re.search(r"v_mfma_f32_16x16x32_bf16", body), name
        meta = asm[asm.index(name + ":"):]
        meta = meta[:meta.index(".end_amdhsa_kernel") + 4000]
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
        assert priv and int(priv[1]) == 0, (name, priv and priv[1])
        vg = re.search(r"; NumVgprs: (\d+)", meta)
        assert vg and int(vg[1]) <= 256, (name, vg and vg[1])
        assert len(re.findall(r"global_load_dwordx4", body)) >= 54, name          # three planes x six taps x three 16-byte pieces


def test_scratch_use_is_bounded(asm):
    for pat, limit in ((r"k_marchILb1ELb0EE", 0), (r"k_marchILb0ELb0EE", 0),
                       (r"k_train_dgrad3ILi8EE", 0), (r"k_train_app3ILi8ELb0EE", 0), (r"k_train_app3ILi8ELb1EE", 0),
                       (r"k_scatter_fixILi8ELb0ELi1024ELi8EE", 0), (r"k_scatter_fixILi24ELb1ELi1024ELi8EE", 0), (r"k_adam_packE", 0)):
        for name, _ in _body(asm, pat):
            meta = asm[asm.index(".amdhsa_kernel " + name):]
            meta = meta[:meta.index(".end_amdhsa_kernel")]
            priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
            assert priv and int(priv[1]) <= limit, (name, priv and priv[1])


def test_colour_backward_kernels_shape(asm):
    """k_train_dgrad3: 96 (W2^T) + 24 (W1^T) MFMAs of the data chain, 4 + 16 selector transposes (feat^T, dz1^T), 24 for dW1.
    k_train_app3: 18 (basis^T), 4 + 12 selector transposes (dfeat^T, X^T per plane), 18 for dbasis.  All on the 32x32x16
    instruction, at most 256 registers (two waves per SIMD)."""
    for pat, n in ((r"k_train_dgrad3ILi8EE", 96 + 24 + 4 + 16 + 24), (r"k_train_app3ILi8ELb0EE", 18 + 4 + 12 + 18), (r"k_train_app3ILi8ELb1EE", 18 + 4 + 12 + 18)):
        name, body = _body(asm, pat)[0]
        assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)) == n, (name, len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)))
        meta = asm[asm.index(name + ":"):]
        meta = meta[:meta.index(".end_amdhsa_kernel") + 4000]
        vg = re.search(r"; NumVgprs: (\d+)", meta)
        assert vg and int(vg[1]) <= 256, (name, vg and vg[1])


WGRAD = ("k_wgrad_w2w3ILi64EE", "k_wgrad_w2w3ILi128EE")


def test_weight_gradient_gemms(asm):
    """k_wgrad_w2w3: split-bf16 on the K = 32 instruction only (the K = 16 form runs at half its rate on gfx950): the
    recomputed layer 1 (8 x 3) + 3 colours x 9 N-tiles x 3 terms per 32-row K-step (the K-step loop is rolled), B
    fragments as ds_read_b128 from the transposed pre-split tile, at most 256 registers (two waves per SIMD).  All of them:
    no scratch, and the staging loads are branch-free, so the wait in front of the LDS stage is a counted vmcnt(n)
    placed by the compiler, not a vmcnt(0) behind a predicated block."""
    def body(kern):
        return _body(asm, kern)[0]
    _, w2 = body("k_wgrad_w2w3ILi128EE")
    assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", w2)) == 24 + 81
    # the double-buffered 64-row form: half a tile's layer 1 (4 x 3) in the prologue and in both orders of the step, the
    # products of a step (2 K-steps, unrolled) in both orders
    _, w64 = body("k_wgrad_w2w3ILi64EE")
    assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", w64)) == 3 * 12 + 2 * 2 * 81
    for t in (w2, w64):
        assert not re.search(r"v_mfma_f32_16x16x16_bf16", t)
        assert len(re.findall(r"ds_read_b128", t)) >= 24
    for kern in WGRAD:
        name, t = body(kern)
        meta = asm[asm.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
        assert priv and int(priv[1]) == 0, (kern, priv and priv[1])
        loads = len(re.findall(r"global_load_dwordx4", t))
        assert loads >= 3, (kern, loads)
        if "Li64E" in kern:           # (its two orders of a step are branches on the wave index by construction: a step's loads are consumed a whole step later)
            continue
        for m in re.finditer(r"s_cbranch_execz (\.LBB\d+_\d+)\n", t):                  # no row load inside a predicated block
            end = t.find("\n" + m[1] + ":", m.end())
            assert end < 0 or "global_load_dwordx4" not in t[m.end():end], (kern, m[1])
