"""The generated ISA of the built library, checked for the one hazard the compiler cannot see: an instruction touching the destination of an
inline-assembly LDS read (ds_read_b64_tr_b16 through ds_read_tr_na) before the hand-placed s_waitcnt lgkmcnt (tools/isa_async_check.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_no_instruction_touches_an_in_flight_lds_read():
    import isa_async_check as chk
    so = os.path.join(ROOT, 'avt_amd', 'libavt_hip.so')
    if not os.path.exists(so):
        pytest.fail('avt_amd/libavt_hip.so is not built (python -c "import __graft_entry__ as g; g.build()")')
    if not os.path.exists(os.path.join(chk.LLVM, 'llvm-objdump')):
        pytest.skip('no llvm-objdump in this image')
    total, n_tr = chk.check_path(so)
    assert n_tr > 1000, n_tr                       # the scan saw the weight-gradient, data-gradient and attention kernels' reads
    assert not total, total


def test_the_checker_flags_a_touch_before_the_wait():
    import isa_async_check as chk
    bad = ['_Z1kv:', '\tds_read_b64_tr_b16 v[4:5], v1 offset:16', '\tv_bfi_b32 v4, s0, v4, v4', '\ts_waitcnt lgkmcnt(0)', '\tv_mov_b32 v9, v5']
    good = ['_Z1kv:', '\tds_read_b64_tr_b16 v[4:5], v1 offset:16', '\tds_read_b64_tr_b16 v[6:7], v1 offset:32', '\ts_waitcnt lgkmcnt(1)',
            '\tv_mov_b32 v9, v5', '\ts_waitcnt lgkmcnt(0)', '\tv_mov_b32 v9, v6']
    late = ['_Z1kv:', '\tds_read_b64_tr_b16 v[4:5], v1 offset:16', '\tds_read_b64_tr_b16 v[6:7], v1 offset:32', '\ts_waitcnt lgkmcnt(1)', '\tv_mov_b32 v9, v7']
    assert chk.check_lines(bad, 'bad')[0] == {'_Z1kv': 1}
    assert chk.check_lines(good, 'good')[0] == {}
    assert chk.check_lines(late, 'late')[0] == {'_Z1kv': 1}
