"""cartpoleplusplus_amd/csrc/tools/check_async_loads.py -- the build's check of conv_k16.h's hand-counted `s_waitcnt vmcnt(N)` (csrc/Makefile, target
check-waits) -- on doctored listings: it must accept a correct loop (loads a whole iteration ahead, two alternative store groups of the
same length on a branch) and reject, deterministically, each way the contract can break: a wait count that is too large, a store that
went missing on one path, a compiler copy of a register whose load is still in flight, a wait that was moved behind its first use."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_async_loads", os.path.join(ROOT, "cartpoleplusplus_amd", "csrc", "tools", "check_async_loads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)
GOOD = open(os.path.join(ROOT, "tests", "golden", "wait_checker_listing.s")).read()


def bad_count(text):
    res = chk.check_listing(text)
    assert list(res) == ["_Z4demov"], list(res)
    return len(res["_Z4demov"])


def test_a_correct_loop_passes():
    assert bad_count(GOOD) == 0


def test_a_wait_count_that_is_too_large_is_caught():
    # chunk 1's wait: the loads of chunk 2 and of the next row's chunk 0 plus the row's two stores = 4 instructions are younger; 5 lets
    # the wait pass while chunk 1's own load is still outstanding
    assert GOOD.count("s_waitcnt vmcnt(4)\n\tv_mfma_f32_16x16x32_f16 v[32:35], v[4:7]") == 1
    bad = GOOD.replace("s_waitcnt vmcnt(4)\n\tv_mfma_f32_16x16x32_f16 v[32:35], v[4:7]", "s_waitcnt vmcnt(5)\n\tv_mfma_f32_16x16x32_f16 v[32:35], v[4:7]")
    assert bad_count(bad) >= 1


def test_a_store_missing_on_one_path_is_caught():
    # the counts assume two stores per row on EVERY path: with one of them gone from the second path the same vmcnt(4) no longer covers
    bad = GOOD.replace("\tbuffer_store_dword v43, v22, s[16:19], 0 offen offset:64\n", "")
    assert bad_count(bad) >= 1
    # ... while an EXTRA store only makes the wait longer than needed: accepted
    more = GOOD.replace("\tbuffer_store_dword v43, v22, s[16:19], 0 offen offset:64\n",
                        "\tbuffer_store_dword v43, v22, s[16:19], 0 offen offset:64\n\tbuffer_store_dword v43, v22, s[16:19], 0 offen offset:128\n")
    assert bad_count(more) == 0


def test_a_copy_of_an_in_flight_register_is_caught():
    # what a register allocator under pressure may do with an inline-asm load's destination: park it before the data has arrived
    bad = GOOD.replace("\ts_add_i32 s20, s20, 1\n", "\tv_mov_b32_e32 v50, v9\n\ts_add_i32 s20, s20, 1\n")
    assert bad_count(bad) >= 1


def test_a_wait_moved_behind_its_first_use_is_caught():
    bad = GOOD.replace("\ts_waitcnt vmcnt(4)\n\tv_mfma_f32_16x16x32_f16 v[32:35], v[8:11], v[24:27], v[32:35]\n",
                       "\tv_mfma_f32_16x16x32_f16 v[32:35], v[8:11], v[24:27], v[32:35]\n\ts_waitcnt vmcnt(4)\n")
    assert bad_count(bad) >= 1


def test_the_build_runs_the_checker():
    mk = open(os.path.join(ROOT, "cartpoleplusplus_amd", "csrc", "Makefile")).read()
    assert "check_async_loads.py" in mk and "check-waits" in mk
    assert "check-waits" in mk.split("all:")[1].split("\n")[0]          # part of `all`
