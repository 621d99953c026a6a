"""tools/check_prefetch_isa.py: the static guard of the inline-asm software prefetch (hipcc does not count asm loads, so
a compiler-generated instruction that touches a register an asm load is still filling reads or destroys data in flight).
Round 4 found a real hazard the original layout-order walk missed -- a wait on a side path that is laid out first -- so
the checker follows the control-flow graph now; these cases pin that, and the product's kernels must pass it."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_prefetch_isa", os.path.join(ROOT, "tools", "check_prefetch_isa.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = "_Z1kv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"
LOAD = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[10:11], v[2:3], off\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def test_copy_before_the_wait_is_flagged():
    v, st = chk.check(HEAD + LOAD + "\tv_mov_b64_e32 v[20:21], v[10:11]\n" + WAIT + TAIL)
    assert len(v) == 1 and v[0][3] == [10, 11]
    assert st["_Z1kv"] == (1, 1)


def test_use_after_the_wait_is_clean():
    v, _ = chk.check(HEAD + LOAD + "\tv_add_f64 v[4:5], v[6:7], v[8:9]\n" + WAIT + "\tv_mov_b64_e32 v[20:21], v[10:11]\n" + TAIL)
    assert v == []


def test_wait_on_a_side_path_laid_out_first_does_not_retire_the_main_path():
    """The round-4 hazard: load; branch to the main path; the side path (with its own wait) is laid out first; the
    main path copies the in-flight register in front of ITS wait. A layout-order walk sees load, wait, copy."""
    text = (HEAD + LOAD +
            "\ts_cbranch_execz .LBB0_2\n"
            "; %bb.1:\n" + WAIT +
            "\ts_branch .LBB0_3\n"
            ".LBB0_2:\n"
            "\tv_mov_b64_e32 v[20:21], v[10:11]\n" + WAIT +
            ".LBB0_3:\n"
            "\tv_mov_b64_e32 v[22:23], v[10:11]\n" + TAIL)
    v, _ = chk.check(text)
    assert [x[2] for x in v] == ["v_mov_b64_e32 v[20:21], v[10:11]"]


def test_loop_carried_load_reaches_the_top_of_the_next_iteration():
    text = (HEAD +
            ".LBB0_1:\n"
            "\tv_mov_b64_e32 v[20:21], v[10:11]\n" + LOAD +
            "\ts_cbranch_scc1 .LBB0_1\n" + WAIT + TAIL)
    v, _ = chk.check(text)
    assert len(v) == 1  # the copy at the loop top runs again with the load of the previous iteration in flight


def test_lds_dma_has_no_register_destination():
    text = HEAD + "\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v[10:11], off\n\t;;#ASMEND\n\tv_mov_b64_e32 v[20:21], v[10:11]\n" + TAIL
    v, st = chk.check(text)
    assert v == [] and st == {}


def test_product_kernels_pass():
    text = chk.listing()
    v, st = chk.check(text)
    assert any("tree_up_micro_kernel" in f for f in st), "no inline-asm prefetch found: nothing was checked"
    assert v == [], v[:5]
