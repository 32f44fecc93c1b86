"""cartpoleplusplus_amd/csrc/tools/check_store_data.py -- the build's check (csrc/Makefile, target check-stores) that no VALU instruction writes
the data registers of a buffer store of more than 64 bits inside the window in which the store still reads them -- on doctored listings.
The hazard is real on gfx950 and outside LLVM's recogniser when the store's soffset is an SGPR (round 6: conv_dx_rs.h at 64-wide rows stored
|x| for x now and then; profiles/NOTEBOOK_r06.md 10); `buffer_store_b128_held` (conv_kyo.h) is the guard the kernels use."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_store_data", os.path.join(ROOT, "cartpoleplusplus_amd", "csrc", "tools", "check_store_data.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def hits(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("_Z4demov:\n" + "".join("\t" + l + "\n" for l in body) + "\ts_endpgm\n")
    return chk.check(str(p))


STORE = "buffer_store_dwordx4 v[200:203], v174, s[0:3], s4 offen"


def test_the_sequence_that_failed_on_the_gpu_is_caught(tmp_path):
    # the tail block of conv_dx_rs_kernel<5, 4, 2> with its row maximum live: address and data registers reused right behind the store
    h = hits(tmp_path, [STORE, "v_max_f32_e64 v174, |v203|, |v203|", "v_max_f32_e64 v202, |v202|, |v202|", "v_max_f32_e32 v174, v202, v174"])
    assert len(h) == 1 and h[0][2] == 1 and "v202" in h[0][4]
    assert len(hits(tmp_path, [STORE, "v_max_f32_e64 v201, |v201|, |v201|"])) == 1          # + 0 wait states


def test_the_guarded_store_passes(tmp_path):
    # buffer_store_b128_held: `s_nop 3` behind the store, the data registers alive across it
    assert hits(tmp_path, [STORE, "s_nop 3", "v_max_f32_e64 v202, |v202|, |v202|"]) == []
    assert len(hits(tmp_path, [STORE, "s_nop 1", "v_max_f32_e64 v202, |v202|, |v202|"])) == 1   # two wait states are LLVM's rule where it has one, not ours
    assert hits(tmp_path, [STORE, "v_mov_b32_e32 v1, v2", "v_mov_b32_e32 v3, v2", "v_mov_b32_e32 v4, v2", "v_mov_b32_e32 v5, v2", "v_mov_b32_e32 v200, v2"]) == []


def test_what_is_not_a_hazard_is_not_reported(tmp_path):
    # 64-bit stores leave with their address (conv_rs16.h's bf16 plane stores are overwritten at + 0 by the hundred, bit-reproducibly)
    assert hits(tmp_path, ["buffer_store_dwordx2 v[80:81], v88, s[20:23], s7 offen", "v_and_b32_e32 v81, 0xffff0000, v71"]) == []
    # an MFMA's or a load's destination lands long after the window
    assert hits(tmp_path, [STORE, "v_mfma_f32_16x16x32_bf16 v[200:203], v[42:45], v[134:137], v[162:165]"]) == []
    assert hits(tmp_path, [STORE, "ds_read_b128 v[200:203], v184"]) == []
    # reads of the data registers are fine; other registers are fine; the walk ends at a branch
    assert hits(tmp_path, [STORE, "v_max_f32_e64 v170, |v203|, |v203|", "v_max3_f32 v187, |v200|, |v201|, v170"]) == []
    assert hits(tmp_path, [STORE, "s_cbranch_scc1 .LBB0_2", "v_mov_b32_e32 v200, 0"]) == []


def test_llvms_own_rule_is_the_bar_where_llvm_applies_it(tmp_path):
    # immediate soffset / global stores: the recogniser keeps VALU writes two wait states away (gfx940+); one would be its bug
    imm = "buffer_store_dwordx4 v[60:63], v7, s[8:11], 0 offen"
    assert hits(tmp_path, [imm, "v_mov_b32_e32 v1, v2", "v_mov_b32_e32 v3, v2", "v_mov_b32_e32 v60, v52"]) == []
    assert len(hits(tmp_path, [imm, "v_mov_b32_e32 v1, v2", "v_mov_b32_e32 v60, v52"])) == 1
    glob = "global_store_dwordx4 v[94:95], v[90:93], off"
    assert hits(tmp_path, [glob, "v_mov_b32_e32 v1, v2", "v_mov_b32_e32 v3, v2", "v_mov_b32_e32 v90, v1"]) == []
    assert len(hits(tmp_path, [glob, "v_mov_b32_e32 v90, v1"])) == 1


def test_the_shipped_kernels_hold_their_row_stores(tmp_path):
    """every user of the raw 128-bit buffer-store builtin in the row-streaming headers goes through buffer_store_b128_held, and the
    Makefile checks those translation units' listings on every build"""
    csrc = os.path.join(ROOT, "cartpoleplusplus_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")) and name != "conv_kyo.h":
            text = open(os.path.join(csrc, name)).read()
            for line in text.splitlines():
                if "__builtin_amdgcn_raw_buffer_store_b128" in line or "__builtin_amdgcn_raw_buffer_store_b96" in line:
                    assert line.rstrip().endswith(", 0, 0);") or ", 0, 0)" in line, (name, line.strip()[:120])     # immediate soffset only
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "check-stores" in mk and "all: $(OUT) $(ABL_OUT) check-waits check-stores" in mk
    for tu in ("conv_dx_rs", "conv_fw_rs", "conv_fwd_rs16", "conv2_bwd_pair", "conv3_bwd_pair"):
        assert tu in mk.split("STORE_TUS =")[1].splitlines()[0]


def test_the_listings_the_build_left_are_clean():
    """`make check-stores` writes the listings of the translation units that issue 128-bit buffer stores next to their objects; whatever is
    there (a tree that was built: the driver's build() step runs before the CPU suite) must hold no wide store with its data registers
    overwritten inside the window -- and the set must be the Makefile's."""
    obj = os.path.join(ROOT, "cartpoleplusplus_amd", "lib", "obj")
    mk = open(os.path.join(ROOT, "cartpoleplusplus_amd", "csrc", "Makefile")).read()
    tus = mk.split("STORE_TUS =")[1].splitlines()[0].split()
    have = [t for t in tus if os.path.exists(os.path.join(obj, t + ".s"))]
    if not have:
        import pytest
        pytest.skip("no listings: the library has not been built in this tree")
    assert sorted(have) == sorted(tus), (have, tus)
    for t in have:
        assert chk.check(os.path.join(obj, t + ".s")) == [], t
