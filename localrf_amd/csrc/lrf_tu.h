// lrf_tu.h -- build configuration of the one translation unit of liblrf_hip.so.
//
// The library is compiled WITHOUT hipcc's SLP vectoriser (-fno-slp-vectorize, __graft_entry__.build).  The vectoriser
// packs adjacent scalar fp32 operations into v_pk_mul/fma/add_f32 with crossed operand selects (op_sel:[0,1]: the low
// result reads the high half of a source).  On MI355X such an instruction returns a WRONG low result in lanes 48..63
// when another wave on the same SIMD issues v_mfma_f32_16x16x32_bf16 back to back (scripts/ubench/pk_mfma.hip: exact
// integer arithmetic, 6 % of the iterations; profiles/r08b_packed_fp32_beside_mfma.md).  That -- not a late MFMA operand
// read -- was the cause of the run-to-run differences rounds 1 and 2 chased (docs/GFX950_FINDINGS.md findings 1, 2, 9, 17): the colour
// kernels interpolate (packed arithmetic after vectorisation) in some waves while others run their MFMA chain.
// tests/test_isa_checks.py rejects any packed fp32 instruction with a crossed select in the shipped ISA.
// (Rounds 1-2 linked two units of this source, one with and one without the vectoriser; both now use the same flags,
// so there is one unit.)
#pragma once
