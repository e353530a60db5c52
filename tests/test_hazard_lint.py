"""The build-time MFMA hazard check (tensorbnn_amd/hazard_lint.py, checked_compile.py), on CPU: one synthetic listing per rule, the rules' numbers
against what the INSTALLED compiler's own hazard recognizer inserts into probe kernels (llc), the repair of a listing, the compile driver, and
the product library / a run-time instantiated library disassembled."""
import os
import re
import subprocess

import pytest

from tensorbnn_amd import hazard_lint as hl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = "0000000000001000 <_Z9k_exampleILi1EEvPf>:\n"


def rules(found):
    return [(f[4].split()[0], f[3], f[5]) for f in found]


def lst(body):
    """a disassembly-style listing: one instruction per line, addresses 8 bytes apart (branches in these tests use labels)"""
    out, addr = [HDR], 0x1000
    for ln in body.strip().split("\n"):
        ln = ln.strip()
        if ln.endswith(":"):
            out.append(ln + "\n")
            continue
        out.append(f"\t{ln:<60}// {addr:012X}: 00000000\n")
        addr += 8
    return "".join(out)


def slist(body):
    """a compiler-style listing (-S): labels, no addresses"""
    return "_Z9k_exampleILi1EEvPf:\n" + "".join(("\t" + l.strip() + "\n") if not l.strip().endswith(":") else l.strip() + "\n" for l in body.strip().split("\n"))


# ---- R1: VALU write -> MFMA operand read (2 wait states), ArchVGPRs and AccVGPRs, SrcA / SrcB / SrcC
def test_r1_valu_write_then_mfma_read_counts_wait_states():
    listing = lst("""
	v_accvgpr_read_b32 v60, a94
	s_waitcnt vmcnt(1)
	v_mfma_f32_16x16x4_f32 a[76:79], v6, v60, a[76:79]
	v_accvgpr_read_b32 v61, a95
	s_nop 1
	v_mfma_f32_16x16x4_f32 a[76:79], v7, v61, a[76:79]
	v_fma_f32 v8, v1, v2, v3
	v_mov_b32_e32 v9, v10
	v_mfma_f32_16x16x4_f32 a[0:3], v8, v62, a[0:3]
	v_mov_b32_e32 v11, v10
	v_mov_b32_e32 v12, v10
	v_mov_b32_e32 v14, v10
	v_mfma_f32_16x16x4_f32 a[0:3], v13, v11, a[0:3]
	s_endpgm
""")
    found = hl.hazards(listing)
    # pair 1: one wait state (the s_waitcnt) between -> hazard; pair 2: s_nop 1 = two -> fine; pair 3: v_fma then one instruction -> hazard;
    # pair 4: two instructions between the write and the MFMA -> fine
    assert [(f[1].split()[0], f[3], f[4].split()[0]) for f in found] == [("v_accvgpr_read_b32", 1, "R1"), ("v_fma_f32", 1, "R1")], found
    assert all(f[0] == "_Z9k_exampleILi1EEvPf" for f in found)
    assert len(hl.hazards(listing, need=3)) == 4
    assert "k_example" in hl.describe(found) and "R1" in hl.describe(found)


def test_r1_sees_accvgpr_operands_and_accvgpr_writers():
    """round 5's N-fringe MFMAs take A, B and C from AccVGPRs, and v_accvgpr_write / _mov are VALU writes like any other"""
    listing = lst("""
	v_accvgpr_write_b32 a7, v1
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a7, a9, a[0:3]
	s_nop 4
	v_accvgpr_mov_b32 a9, a20
	s_nop 0
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a8, a9, a[0:3]
	s_nop 4
	v_accvgpr_write_b32 a2, v1
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a8, a10, a[0:3]
	s_nop 4
	v_accvgpr_write_b32 a12, v1
	s_nop 1
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a12, a10, a[0:3]
	s_endpgm
""")
    found = [f for f in hl.hazards_cfg(listing) if f[4].startswith("R1")]
    assert [(f[1].split()[1].rstrip(","), f[3]) for f in found] == [("a7", 0), ("a9", 1), ("a2", 0)], found      # SrcA, SrcB, SrcC; the last pair is fine


def test_operand_modifiers_do_not_hide_a_register():
    assert hl._regs("-v1") == {1} and hl._regs("|v7|") == {7} and hl._regs("neg(v3)") == {3} and hl._regs("sext(v4)") == {4}
    assert hl._regs("v[4:5] dst_sel:DWORD") == {4, 5} and hl._regs("a[2:3]") == {hl.AOFF + 2, hl.AOFF + 3}
    assert hl._regs("vcc") == set() and hl._regs("s[4:5]") == set() and hl._regs("0x3f800000") == set() and hl._regs("exec") == set()
    listing = lst("""
	v_mul_f32_e64 v5, -v1, |v2|
	v_mfma_f32_16x16x4_f32 a[0:3], v5, v6, a[0:3]
	s_nop 9
	v_add_f32_e64 v9, -v1, |a1|
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R1", 0, 2)]


# ---- R2: an MFMA result touched before it has landed
def test_r2a_mfma_result_as_srca_or_srcb_of_the_next_mfma():
    listing = lst("""
	v_mfma_f32_16x16x4_f32 v[0:3], v10, v11, 0
	s_nop 8
	v_mfma_f32_16x16x4_f32 v[4:7], v0, v11, 0
	v_mfma_f32_4x4x1_16b_f32 a[0:3], v10, v11, 0
	s_nop 2
	v_mfma_f32_4x4x1_16b_f32 a[4:7], v10, a3, 0
	v_mfma_f32_32x32x2_f32 a[16:31], v10, v11, 0
	s_nop 15
	s_nop 1
	v_mfma_f32_16x16x4_f32 a[8:11], a31, v11, 0
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R2a", 9, 10), ("R2a", 3, 4)]      # 16 passes: 18 needed, 18 there


def test_r2b_accumulation_into_the_predecessors_result():
    """the same registers back to back: free after an 8-pass MFMA, two wait states after a 2-pass one (round 5: a 4x4x1 accumulating into its
    predecessor's result read a stale accumulator); overlapping but different registers: the producer's passes"""
    listing = lst("""
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, a[0:3]
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, a[0:3]
	v_mfma_f32_4x4x1_16b_f32 a[4:7], v10, v11, a[4:7]
	s_nop 0
	v_mfma_f32_4x4x1_16b_f32 a[4:7], v10, v11, a[4:7]
	s_nop 1
	v_mfma_f32_4x4x1_16b_f32 a[4:7], v10, v11, a[4:7]
	s_nop 9
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, a[0:3]
	s_nop 6
	v_mfma_f32_16x16x4_f32 a[8:11], v10, v11, a[2:5]
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R2b", 1, 2), ("R2b", 7, 8)]


def test_r2c_valu_reads_or_overwrites_an_mfma_result():
    """the read (round 5: an inline-asm v_pk_mul ahead of mfma_settle) and the write (round 6: an asm v_pk_mul allocated to the unused
    registers of a lane-sum MFMA, one instruction behind it)"""
    listing = lst("""
	v_mfma_f32_16x16x4_f32 v[38:41], v151, v50, 0
	v_pk_mul_f32 v[40:41], v[106:107], v[98:99] clamp
	s_nop 9
	v_mfma_f32_16x16x4_f32 a[0:3], v151, v50, a[0:3]
	s_nop 8
	v_accvgpr_read_b32 v5, a2
	s_nop 9
	v_mfma_f32_4x4x1_16b_f32 v[52:55], v120, v88, v[58:61]
	s_nop 2
	v_add_f32_e64 v70, v50, v52
	v_mfma_f32_16x16x4_f32 a[0:3], v151, v50, a[0:3]
	s_nop 9
	v_accvgpr_read_b32 v5, a2
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R2c", 0, 10), ("R2c", 9, 10), ("R2c", 3, 4)]


def test_r2d_mfma_result_as_data_or_address_of_a_memory_instruction():
    listing = lst("""
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, 0
	s_nop 8
	ds_write_b128 v0, a[0:3]
	s_nop 9
	v_mfma_f32_16x16x4_f32 v[0:3], v10, v11, 0
	s_nop 5
	global_store_dwordx4 v8, v[0:3], s[0:1]
	s_nop 9
	v_mfma_f32_16x16x4_f32 v[0:3], v10, v11, 0
	s_nop 7
	buffer_load_dword v9, v2, s[0:3], 0 offen
	s_nop 9
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, 0
	s_nop 9
	buffer_store_dwordx4 a[0:3], v8, s[0:3], 0 offen
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R2d", 9, 10), ("R2d", 6, 10), ("R2d", 8, 10)]


def test_r3_exec_write_then_mfma():
    listing = lst("""
	v_cmpx_gt_u32_e32 13, v0
	s_nop 2
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, 0
	s_nop 9
	v_cmpx_gt_u32_e32 13, v0
	s_nop 3
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, a[0:3]
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R3", 3, 4)]


def test_r5_transcendental_result_read_by_the_next_valu_instruction():
    listing = lst("""
	v_exp_f32_e32 v0, v1
	v_pk_mul_f32 v[4:5], v[0:1], v[2:3]
	v_rcp_f32_e32 v6, v7
	s_nop 0
	v_mul_f32_e32 v8, v6, v9
	v_exp_f32_e32 v0, v1
	v_rcp_f32_e32 v10, v0
	v_log_f32_e32 v11, v12
	v_mov_b32_e32 v20, v21
	v_add_f32_e32 v13, v11, v11
	s_endpgm
""")
    # the packed multiply right behind v_exp reads its result: hazard; behind s_nop 0: fine; a transcendental reader: fine; one instruction between: fine
    assert rules(hl.hazards_cfg(listing)) == [("R5", 0, 1)]
    new, fixed, left = hl.fix_listing(slist("""
	v_exp_f32_e32 v0, v1
	v_pk_mul_f32 v[4:5], v[0:1], v[2:3]
	s_endpgm
"""))
    assert left == [] and "s_nop 0" in new and hl.hazards_cfg(new) == []


# ---- R4: a loaded register touched before a wait that covers the load
def test_r4_register_of_an_asm_load_touched_before_the_wait():
    """nf_load's `ds_read_b128 a[..]` is inline asm: the compiler does not count it, and a copy the register allocator puts between the load
    and nf_mfma's own s_waitcnt would read the register early"""
    listing = lst("""
	ds_read_b128 a[4:7], v1
	ds_read_b128 a[8:11], v1 offset:16
	v_accvgpr_read_b32 v9, a5
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a4, a8, a[0:3]
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R4", 1, 0)]
    ok = lst("""
	ds_read_b128 a[4:7], v1
	ds_read_b128 a[8:11], v1 offset:16
	s_load_dword s4, s[0:1], 0x0
	s_waitcnt lgkmcnt(1)
	v_accvgpr_read_b32 v9, a5
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_4x4x1_16b_f32 a[0:3], a4, a8, a[0:3]
	s_endpgm
""")
    # lgkmcnt(1) with ONE later LDS operation covers the first load (LDS returns in order; the scalar load may overtake and is not counted)
    assert hl.hazards_cfg(ok) == []
    bad = ok.replace("ds_read_b128 a[8:11], v1 offset:16", "s_load_dword s5, s[0:1], 0x4     ")
    assert rules(hl.hazards_cfg(bad)) == [("R4", 0, 0)]            # ... but with only scalar loads behind it, lgkmcnt(1) proves nothing


def test_r4_vector_memory_counter_and_loop_carried_loads():
    listing = slist("""
.LBB0_1:
	buffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen
	buffer_load_dwordx4 v[8:11], v1, s[0:3], 0 offen offset:16
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v20, v4, v5
	v_add_f32_e32 v21, v8, v9
	s_waitcnt vmcnt(0)
	s_cbranch_scc1 .LBB0_1
	s_endpgm
""")
    assert rules(hl.hazards_cfg(listing)) == [("R4", 0, 0)]         # v8 read under vmcnt(1): only the first load is covered
    # an LDS-DMA load names an ADDRESS register, not a destination
    dma = lst("""
	buffer_load_dword v1, s[0:3], 0 offen lds
	v_add_u32_e32 v1, 64, v1
	s_waitcnt vmcnt(0)
	s_endpgm
""")
    assert hl.hazards_cfg(dma) == []


def test_r4_a_call_waits_for_everything():
    """a function that is not a kernel opens with s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0): a load issued before a call has landed behind it
    (k_fwd_bwd_tall with four dense layers calls TallCfg::boff at run time)"""
    listing = lst("""
	global_load_dword v181, v[0:1], off
	s_swappc_b64 s[30:31], s[0:1]
	v_sub_f32_e32 v181, v181, v182
	s_endpgm
""")
    assert hl.hazards_cfg(listing) == []
    assert rules(hl.hazards_cfg(listing.replace("s_swappc_b64 s[30:31], s[0:1]", "s_nop 0                      "))) == [("R4", 0, 0)]


# ---- control flow
def test_pairs_are_followed_across_branches_and_joins():
    """the pair the compiler's own hazard recognizer missed (cooperative tail of the narrow kernel, round 5): an MFMA ends a wave-uniform block,
    two branches later a move at the join reads its result"""
    listing = """
0000000000001000 <_Z9k_exampleILi2EEvPf>:
	v_mfma_f32_16x16x4_f32 v[52:55], v55, v51, v[146:149]      // 000000001000: D3C58034
	s_cbranch_execz 1                                          // 000000001008: BF880001
	s_branch 3                                                 // 00000000100C: BF820003
	v_mov_b32_e32 v1, v2                                       // 000000001010: 7E020302
	v_mov_b32_e32 v3, v2                                       // 000000001014: 7E060302
	v_mov_b32_e32 v4, v2                                       // 000000001018: 7E080302
	v_mov_b64_e32 v[90:91], v[54:55]                           // 00000000101C: 7EB47136
	s_nop 9                                                    // 000000001020: BF800009
	v_mov_b64_e32 v[88:89], v[52:53]                           // 000000001024: 7EB07134
	s_endpgm                                                   // 000000001028: BF810000
"""
    found = hl.hazards_cfg(listing)
    # through `s_branch 3` (target 0x101C) the move reads v[54:55] two wait states after the MFMA; the second move sits behind s_nop 9: fine
    assert [(f[2].split()[0], f[2].split()[1].rstrip(","), f[3], f[4].split()[0]) for f in found] == [("v_mov_b64_e32", "v[90:91]", 2, "R2c")], found
    assert hl.hazards(listing) == []                         # the straight-line reading stops at the branches
    # the branch target comes from the ENCODED word: a symbolized operand reads the same
    sym = listing.replace("s_branch 3       ", "s_branch <k+0x1c>")
    assert len(hl.hazards_cfg(sym)) == 1


def test_a_branch_that_cannot_be_followed_is_a_finding():
    listing = """
0000000000001000 <_Z9k_exampleILi3EEvPf>:
	s_cbranch_scc1 200                                         // 000000001000: BF8500C8
	s_setpc_b64 s[4:5]                                         // 000000001004: BE801D04
"""
    assert sorted(r[0] for r in rules(hl.hazards_cfg(listing))) == ["R0", "R0"]
    ret = """
0000000000001000 <_Z4offWi>:
	v_mov_b32_e32 v0, 2                                        // 000000001000: 7E000282
	s_setpc_b64 s[30:31]                                       // 000000001004: BE801D1E
"""
    assert hl.hazards_cfg(ret) == []                         # a function's return


def test_compiler_listing_with_labels_and_a_relaxed_long_branch():
    listing = slist("""
	v_mfma_f32_16x16x4_f32 a[0:3], v10, v11, a[0:3]
	s_cbranch_vccz .LBB0_2
	s_getpc_b64 s[4:5]
.Lpost_getpc0:
	s_add_u32 s4, s4, (.LBB0_3-.Lpost_getpc0)&4294967295
	s_addc_u32 s5, s5, (.LBB0_3-.Lpost_getpc0)>>32
	s_setpc_b64 s[4:5]
.LBB0_2:
	s_nop 9
	v_accvgpr_read_b32 v1, a0
	s_endpgm
.LBB0_3:
	v_accvgpr_read_b32 v2, a1
	s_endpgm
""")
    found = hl.hazards_cfg(listing)
    assert rules(found) == [("R2c", 5, 10)] and found[0][2].startswith("v_accvgpr_read_b32 v2")


# ---- the repair
def test_fix_listing_inserts_exactly_the_missing_wait_states():
    listing = slist("""
	v_mfma_f32_16x16x4_f32 v[38:41], v151, v50, 0
	v_pk_mul_f32 v[40:41], v[106:107], v[98:99] clamp
	v_accvgpr_read_b32 v60, a94
	v_mfma_f32_16x16x4_f32 a[76:79], v6, v60, a[76:79]
	s_cbranch_vccz .LBB0_2
	s_nop 3
.LBB0_2:
	v_accvgpr_read_b32 v1, a77
	ds_read_b128 a[4:7], v1
	v_accvgpr_read_b32 v9, a5
	v_mfma_f32_32x32x2_f32 a[16:31], v10, v11, 0
	global_store_dwordx4 v8, a[16:19], s[0:1]
	s_endpgm
""")
    new, fixed, left = hl.fix_listing(listing)
    assert left == [] and hl.hazards_cfg(new) == []
    assert sorted({f[4].split()[0] for f in fixed}) == ["R1", "R2c", "R2d", "R4"]
    added = [l.strip() for l in new.split("\n") if l not in listing.split("\n")]
    assert added == ["s_nop 9", "s_nop 1", "s_nop 8", "s_waitcnt lgkmcnt(0)", "s_nop 15", "s_nop 1"], added
    # nothing to do: the same object comes back
    clean, fixed2, left2 = hl.fix_listing(new)
    assert clean is new and fixed2 == [] and left2 == []


# ---- the numbers, against the installed compiler's own hazard recognizer
PROBES = """
declare <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float, float, <4 x float>, i32, i32, i32)
declare <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float, float, <4 x float>, i32, i32, i32)
declare <16 x float> @llvm.amdgcn.mfma.f32.32x32x2f32(float, float, <16 x float>, i32, i32, i32)
define amdgpu_kernel void @r2a_16(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  %e = extractelement <4 x float> %c, i32 0
  %d = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %e, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %d, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2a_4(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  %e = extractelement <4 x float> %c, i32 0
  %d = call <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float %e, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %d, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2b_4(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  %d = call <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float %a, float %b, <4 x float> %c, i32 0, i32 0, i32 0)
  store <4 x float> %d, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2b_16(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  %d = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> %c, i32 0, i32 0, i32 0)
  store <4 x float> %d, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2c_16(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  %e = fadd <4 x float> %c, <float 1.0, float 1.0, float 1.0, float 1.0>
  store <4 x float> %e, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2d_16(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %c, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2d_4(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.4x4x1f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %c, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2d_32(ptr addrspace(1) %p, float %a, float %b) {
  %c = call <16 x float> @llvm.amdgcn.mfma.f32.32x32x2f32(float %a, float %b, <16 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <16 x float> %c, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r2d_lds(ptr addrspace(3) %l, float %a, float %b) {
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %a, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %c, ptr addrspace(3) %l
  ret void
}
declare float @llvm.amdgcn.exp2.f32(float)
define amdgpu_kernel void @r5(ptr addrspace(1) %p, float %a) {
  %e = call float @llvm.amdgcn.exp2.f32(float %a)
  %f = fadd float %e, 1.0
  store float %f, ptr addrspace(1) %p
  ret void
}
define amdgpu_kernel void @r1(ptr addrspace(1) %p, float %a, float %b) {
  %x = fadd float %a, %b
  %c = call <4 x float> @llvm.amdgcn.mfma.f32.16x16x4f32(float %x, float %b, <4 x float> zeroinitializer, i32 0, i32 0, i32 0)
  store <4 x float> %c, ptr addrspace(1) %p
  ret void
}
"""
# probe -> (rule the probe exercises, wait states the rule says the pair needs)
EXPECT = {"r2a_16": ("R2a", 10), "r2a_4": ("R2a", 4), "r2b_4": ("R2b", 2), "r2b_16": (None, 0), "r2c_16": ("R2c", 10), "r2d_16": ("R2d", 10),
          "r2d_4": ("R2d", 4), "r2d_32": ("R2d", 18), "r2d_lds": ("R2d", 10), "r1": ("R1", 2), "r5": ("R5", 1)}


def test_hazard_rules_agree_with_the_installed_compiler(tmp_path):
    """llc (the compiler hipcc drives) compiles one probe kernel per rule: its hazard recognizer puts `s_nop`s between the pair.  (a) the check
    finds nothing in the compiler's output; (b) with the `s_nop`s taken out, the check finds the pair under the expected rule -- and (c) putting
    back ONE wait state fewer than the rule's number still shows it, the rule's number exactly does not: the numbers in hazard_lint.py are the
    numbers of the GCNHazardRecognizer that is installed here."""
    llc = os.path.join(hl.LLVM_BIN, "llc")
    if not os.path.exists(llc):
        pytest.skip("no llc next to the compiler")
    src = tmp_path / "probes.ll"
    src.write_text(PROBES)
    out = tmp_path / "probes.s"
    subprocess.run([llc, "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", str(src), "-o", str(out)], check=True)
    text = out.read_text()
    assert hl.hazards_cfg(text) == []                                                        # (a)
    funcs = {}
    for name in EXPECT:
        body = text[text.index(f"\n{name}:"):text.index(f".Lfunc_end", text.index(f"\n{name}:"))]
        funcs[name] = body
    for name, (rule, need) in EXPECT.items():
        body = funcs[name]
        lines = body.split("\n")
        imf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        # the window this probe is about: between its two MFMAs, or between its MFMA and the consumer behind it; for r1 in front of the MFMA
        if rule == "R1":
            lo, hi = 0, imf[0]
        elif rule == "R5":
            lo, hi = 0, len(lines)                         # (no MFMA in this probe: the transcendental and its reader)
        elif len(imf) == 2:
            lo, hi = imf[0], imf[1]
        else:
            lo, hi = imf[0], len(lines)
        stripped = [l for i, l in enumerate(lines) if not (lo < i < hi and l.strip().startswith("s_nop"))]
        found = [f for f in hl.hazards_cfg("\n".join(stripped)) if rule and f[4].startswith(rule)]
        if rule is None:
            assert not any(l.strip().startswith("s_nop") for l in lines[lo + 1:hi]), name       # the compiler needs nothing either
            continue
        assert found and found[0][5] == need, (name, found)                                     # (b)
        found.sort(key=lambda f: f[3])
        seen = found[0][3]                                                                      # (the closest producer of the window)
        # (c) the consumer's line, with need - seen - 1 and need - seen wait states in front of it
        cons = found[0][2]
        ic = max(i for i, l in enumerate(stripped) if l.strip().split(";")[0].strip() == cons)
        for extra, clean in ((need - seen - 1, False), (need - seen, True)):
            trial = list(stripped)
            if extra > 0:
                trial[ic:ic] = [f"\ts_nop {extra - 1}"]
            got = [f for f in hl.hazards_cfg("\n".join(trial)) if f[4].startswith(rule)]
            assert (got == []) == clean, (name, extra, got)
        # and the compiler's own distance is the rule's number: never more than one instruction of slack (it counts an unrelated instruction
        # in the window as a wait state, as the check does)
        comp = [f for f in hl.hazards_cfg(body, need=need) if f[4].startswith(rule)] if rule == "R1" else []
        assert comp == []


# ---- the compile driver and the libraries it produced
def _tools():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return os.path.exists(hipcc) and hl.available()


def test_built_library_has_no_mfma_hazard():
    """the product library's device code, disassembled: what checked_compile verified object by object when build.py built it"""
    from tensorbnn_amd import _native as nat
    st = nat.lint_status()
    assert all(u in st for u in ("tbnn_api.hip", "tbnn_mid.hip", "tbnn_tall.hip", "tbnn_wide.hip")) and "listing checked" in st
    if not hl.available():
        pytest.skip("no llvm-objdump on this machine")
    text = hl.disassemble(nat.LIB_PATH)
    assert text.count("v_mfma_f32_16x16x4") > 1000           # the kernels are in there
    assert hl.hazards_cfg(text) == []


def test_checked_compile_repairs_what_the_plain_compile_leaves(tmp_path, monkeypatch):
    """15 -> 170 -> 114 -> 1 on the wide family: with two waves per SIMD the register allocator parks a-blocks of k_dw_wide in AccVGPRs and brings
    one back (v_accvgpr_read) straight in front of the inline-asm MFMA that reads it (round 5: wrong, unrepeatable dW tiles).  Since round 6 every
    asm MFMA carries its own two wait states (TBNN_ASM_MFMA_NOP=1, the default): jit.build hands out a library without any finding.  The same
    source compiled plainly WITHOUT them, and through the checked compile: the latter repairs what the former shows (a pair that is certainly
    left by a plain compile: test_checked_compile_needs_no_disassembler).  No GPU needed: hipcc cross-compiles, llvm-objdump disassembles."""
    from tensorbnn_amd import jit, checked_compile as cc, _native as nat
    if not _tools():
        pytest.skip("needs hipcc and llvm-objdump")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    monkeypatch.setenv("TBNN_JIT_DIR", str(tmp_path))
    monkeypatch.setenv("TBNN_JIT_SKIP", "fast3,fast,mid,tall")
    dims = [15, 170, 114, 1]
    layers = [(dims[i], dims[i + 1], nat.ACT_RELU if i < len(dims) - 2 else nat.ACT_NONE, nat.PRIOR_CAUCHY) for i in range(len(dims) - 1)]
    so = jit.build(layers, nat.LIK_GAUSSIAN)
    assert so and os.path.exists(so)
    assert hl.check(so) == []
    st = jit.lint_status(so)
    assert st.startswith("wide:") and "listing checked" in st and "disassembly clean" in st, st
    # the same translation unit without the wait states in the asm statements, compiled plainly
    src = tmp_path / "plain.hip"
    src.write_text(jit.source(dims, nat.ACT_RELU, nat.ACT_NONE, False, "wide"))
    plain = tmp_path / "plain.so"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-DTBNN_ASM_MFMA_NOP=0"] + jit.NARROW_FLAGS
    subprocess.run(cmd + ["-o", str(plain), str(src)], check=True, stderr=subprocess.DEVNULL)
    found = hl.check(str(plain))
    # (round 5: `v_accvgpr_read` of a parked a-block straight in front of the asm MFMA, R1.  Round 6 gave the accumulators their own register class
    # from birth -- acc_zero -- and the register pressure behind the parking went with it: the pair may be gone from this shape; whatever the plain
    # compile shows, the checked compile of the same source must have repaired exactly when there was something to repair)
    assert all(f[4].split()[0] in ("R1", "R2a", "R2b", "R2c", "R2d", "R3", "R4") for f in found), hl.describe(found)
    fixed = tmp_path / "fixed.so"
    r = cc.run(cmd + ["-o", str(fixed), str(src)])
    m = re.search(r"listing checked \((\d+) repaired", r.status)
    assert r.rc == 0 and m and "disassembly clean" in r.status, r.status
    assert (int(m.group(1)) > 0) == bool(found), (r.status, hl.describe(found))
    assert hl.check(str(fixed)) == []


def test_checked_compile_needs_no_disassembler(tmp_path, monkeypatch):
    """the check runs on the compiler's own listing: without llvm-objdump the unit is still checked (and says that the second look was skipped)"""
    from tensorbnn_amd import checked_compile as cc
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    src = tmp_path / "k.hip"
    src.write_text("""
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* p, float a, float b) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    float d;
    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(d) : "v"(c[0]));      // an inline-asm reader of a raw MFMA result: no wait states from the compiler
    p[threadIdx.x] = d;
}
""")
    monkeypatch.setattr(hl, "LLVM_BIN", str(tmp_path / "nowhere"))
    out = tmp_path / "k.o"
    r = cc.run([hipcc, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(out)], keep_listing=str(tmp_path / "k.s"))
    assert r.rc == 0 and out.exists(), r.stderr[-500:]
    assert "1 repaired: R2c:1" in r.status and "disassembly" not in r.status, r.status
    fixed = (tmp_path / "k.s").read_text()
    i = fixed.index("v_mfma_f32_16x16x4_f32")
    assert re.search(r"s_nop \d+\n(\s*;[^\n]*\n)*\s*v_mul_f32", fixed[i:]), fixed[i:i + 400]
    assert hl.hazards_cfg(fixed) == []


def test_checked_compile_with_a_driver_that_prints_no_subcommands(tmp_path, monkeypatch):
    """another toolchain may not answer `-###` the way this hipcc does: the unit is then compiled plainly with the wait states inside every asm MFMA
    and must pass the disassembly check -- clean, or no object; never an unchecked one"""
    from tensorbnn_amd import checked_compile as cc
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not _tools():
        pytest.skip("needs hipcc and llvm-objdump")
    wrapper = tmp_path / "hipcc_quiet"
    wrapper.write_text(f"#!/bin/sh\nfor a in \"$@\"; do [ \"$a\" = \"-###\" ] && exit 0; done\nexec {hipcc} \"$@\"\n")
    wrapper.chmod(0o755)
    src = tmp_path / "k.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float* p) { p[threadIdx.x] = 1.f; }\n")
    out = tmp_path / "k.o"
    r = cc.run([str(wrapper), "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(out)])
    assert r.rc == 0 and out.exists() and "compiled plainly" in r.status and "disassembly clean" in r.status, (r.status, r.stderr[-300:])
    # ... and without a disassembler there is nothing to vouch for it
    monkeypatch.setattr(hl, "LLVM_BIN", str(tmp_path / "nowhere"))
    r = cc.run([str(wrapper), "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(tmp_path / "k2.o")])
    assert r.rc != 0 and "cannot be checked" in r.stderr
