"""The generated ISA of the built library, checked for the one hazard the compiler cannot see: an instruction touching the destination of an
inline-assembly LDS read (ds_read_b64_tr_b16 through ds_read_tr_na) before the hand-placed s_waitcnt lgkmcnt, or the destination of one of the
attention kernels' hand-written global loads (strip_ld_na: requested in one trip of the item loop, retired by a counted s_waitcnt vmcnt at the top of
the next) before that wait (tools/isa_async_check.py)."""
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


def test_no_instruction_touches_an_in_flight_global_load_of_the_attention_kernels():
    import isa_async_check as chk
    so = os.path.join(ROOT, 'avt_amd', 'libavt_hip.so')
    if not os.path.exists(so):
        pytest.fail('avt_amd/libavt_hip.so is not built (python -c "import __graft_entry__ as g; g.build()")')
    if not os.path.exists(os.path.join(chk.LLVM, 'llvm-objdump')):
        pytest.skip('no llvm-objdump in this image')
    total, n_ld = chk.check_path_vmem(so)
    assert n_ld > 200, n_ld                        # forward (Q strip) and backward (K, V, dO, O strips, per-row scalars) of every instantiation
    assert not total, total


def test_the_vmem_checker_follows_the_back_edge():
    import isa_async_check as chk
    # objdump form: text // address: encoding.  Loop top at 0x10; the load in the tail is retired by the counted wait at the top of the next trip.
    def prog(top, tail_extra=()):
        lines = ['0000000000000000 <_Z19vit_attn_fwd_kernelv>:',
                 '\ts_nop 0                          // 000000000000: BF800000',
                 '\ts_nop 0                          // 000000000004: BF800000',
                 '\ts_nop 0                          // 000000000008: BF800000',
                 '\ts_nop 0                          // 00000000000C: BF800000']
        addr = 0x10
        for t in list(top) + ['buffer_load_dwordx4 v[4:7], v2, s[4:7], 0 offen', 'global_store_dwordx4 v[20:21], v[12:15], off'] + list(tail_extra):
            lines.append(f'\t{t}   // {addr:012X}: 00000000')
            addr += 4
        back = (0x10 - (addr + 4)) // 4 + 65536
        lines.append(f'\ts_branch {back}   // {addr:012X}: BF820000')
        lines.append(f'\ts_endpgm   // {addr + 4:012X}: BF810000')
        return lines
    good = prog(['s_waitcnt vmcnt(1)', 'v_mov_b32_e32 v9, v5'])
    copy_before_wait = prog(['v_mov_b32_e32 v9, v5', 's_waitcnt vmcnt(1)'])          # what hipcc did with a second register set (late round 5)
    wait_too_weak = prog(['s_waitcnt vmcnt(2)', 'v_mov_b32_e32 v9, v5'])
    touched_in_the_tail = prog(['s_waitcnt vmcnt(1)'], tail_extra=['v_add_u32_e32 v4, v4, v1'])
    assert chk.check_vmem_lines(good, 'good') == ({}, 3)                # the load is seen once per trip walked: first pass + the back edge twice
    assert chk.check_vmem_lines(copy_before_wait, 'copy')[0].get('_Z19vit_attn_fwd_kernelv', 0) >= 1
    assert chk.check_vmem_lines(wait_too_weak, 'weak')[0].get('_Z19vit_attn_fwd_kernelv', 0) >= 1
    assert chk.check_vmem_lines(touched_in_the_tail, 'tail')[0].get('_Z19vit_attn_fwd_kernelv', 0) >= 1      # (seen once per trip walked)
