"""Static check of the generated ISA for the one thing the compiler cannot know about the inline-assembly LDS reads (ds_read_b64_tr_b16 issued
through `ds_read_tr_na`, avt_amd/csrc/common.hpp): their destination registers are not valid until the hand-placed `s_waitcnt lgkmcnt`.
Flags every instruction that reads or writes such a register between the read and the `s_waitcnt lgkmcnt(N)` that retires it (the LDS
operations of a wave return in order, so a wait for N outstanding retires all but the youngest N).

    python tools/isa_async_check.py [-v] avt_amd/libavt_hip.so        # the device code objects inside the built library (llvm-objdump)
    python tools/isa_async_check.py [-v] file.s                        # hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o file.s file.hip

Found with it in round 3: a union of the two 64-bit halves of a fragment through 16-bit element vectors made hipcc "merge" them with one
`v_bfi_b32 d, 0xffff, s, s` per register BEFORE the wait (32 per K-loop trip of the weight-gradient kernel; a race in a small attention
instantiation).  Round 5: the same walk for the hand-written global loads of the attention kernels (check_vmem_lines).
tests/test_isa_cpu.py runs both on the built library."""
import os, re, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def _regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def check_lines(lines, name, verbose=False, out=None):
    """-> {kernel: count of hazardous instructions}; also counts the transposing reads seen (to prove the scan looked at something)."""
    per_kernel, n_tr = {}, 0
    kernel, pending = None, []        # pending: (destination registers, line number) of LDS / scalar-memory operations in flight, oldest first
    for no, line in enumerate(lines, 1):
        m = re.match(r'^(?:[0-9a-f]+ <)?(_Z\w+)>?:', line)
        if m:
            kernel, pending = m.group(1), []
            continue
        t = re.split(r';|//', line)[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        op = t.split()[0]
        args = [a.strip() for a in t[len(op):].split(',')]
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            continue
        used = set()
        for a in args:
            used |= _regs(a.split()[0] if a else '')
        if op.startswith('ds_') or op.startswith('s_load') or op.startswith('s_buffer_load'):
            if op == 'ds_read_b64_tr_b16':
                n_tr += 1
            pending.append((_regs(args[0]) if op == 'ds_read_b64_tr_b16' else set(), no))
            continue
        for d, at in pending:
            if d & used:
                per_kernel[kernel] = per_kernel.get(kernel, 0) + 1
                if verbose:
                    print(f'{name}:{no}: {kernel}: `{t}` touches v{sorted(d & used)} of the transposing read at line {at} before its wait', file=out or sys.stdout)
    return per_kernel, n_tr


VMEM_KERNELS = re.compile(r'vit_attn_(fwd|bwd1)_kernel')


def check_vmem_lines(lines, name, verbose=False, out=None, kernels=VMEM_KERNELS):
    """The same hazard for the hand-written global loads of the attention kernels (strip_ld_na / dword_ld_na in vit_attention.hip: the next item's
    strips are requested in one trip of the item loop and retired by a counted `s_waitcnt vmcnt(N)` at the top of the next): no instruction may touch
    a destination register between the request and the wait that retires it.  Vector-memory operations retire in order, loads and stores alike; the
    walk is linear and follows every backward `s_branch` twice (the item loop's back edge), i.e. it assumes -- as the counted waits do -- that every
    guarded memory operation on the way is issued.  -> ({kernel: hazardous instructions}, loads with a destination seen)"""
    per_kernel, n_ld = {}, 0
    # split into kernels: [(name, [(addr, text)])]
    funcs, cur = [], None
    for line in lines:
        m = re.match(r'^(?:[0-9a-f]+ <)?(_Z\w+)>?:', line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        if cur is None:
            continue
        parts = re.split(r';|//', line)
        t = parts[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        am = re.match(r'\s*([0-9A-Fa-f]+):', parts[1]) if len(parts) > 1 else None
        cur[1].append((int(am.group(1), 16) if am else None, t))
    for kernel, ins in funcs:
        if not kernels.search(kernel):
            continue
        index = {a: i for i, (a, _) in enumerate(ins) if a is not None}
        pending, followed, i, steps = [], {}, 0, 0
        while i < len(ins) and steps < 8 * len(ins):
            steps += 1
            addr, t = ins[i]
            op = t.split()[0]
            args = [a.strip() for a in t[len(op):].split(',')]
            if op == 's_endpgm':
                break
            if op == 's_branch' and addr is not None:
                off = int(args[0])
                off = off - 65536 if off >= 32768 else off
                tgt = addr + 4 + 4 * off
                if off < 0 and followed.get(i, 0) < 2 and tgt in index:
                    followed[i] = followed.get(i, 0) + 1
                    i = index[tgt]
                    continue
                i += 1                         # (a back edge taken twice already: fall through to whatever lies behind the loop)
                continue
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', t)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
                i += 1
                continue
            is_vmem = op.startswith(('buffer_', 'global_', 'scratch_', 'flat_'))
            is_load = is_vmem and ('_load' in op) and ' lds' not in t and not t.endswith('lds')
            srcs = args[1:] if is_load else args
            used = set()
            for a in srcs:
                used |= _regs(a.split()[0] if a else '')
            if not is_vmem:
                used |= _regs(args[0].split()[0]) if args and args[0] else set()
            for d, at in pending:
                if d & used:
                    per_kernel[kernel] = per_kernel.get(kernel, 0) + 1
                    if verbose:
                        print(f'{name}: {kernel}: `{t}` touches v{sorted(d & used)} of the load `{ins[at][1]}` before its wait', file=out or sys.stdout)
            if is_vmem:
                dest = _regs(args[0].split()[0]) if is_load else set()
                n_ld += 1 if dest else 0
                pending.append((dest, i))
            i += 1
    return per_kernel, n_ld


def device_disassembly(so_path):
    """The gfx950 code objects of a HIP shared library, disassembled: [(name, [lines])]."""
    res = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run([f'{LLVM}/llvm-objcopy', '-O', 'binary', '--only-section=.hip_fatbin', so_path, fat], check=True)
        data = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        pos = [m.start() for m in re.finditer(re.escape(magic), data)]
        for i, p in enumerate(pos):
            b, o = os.path.join(td, f'b{i}.bin'), os.path.join(td, f'co{i}.o')
            open(b, 'wb').write(data[p:pos[i + 1] if i + 1 < len(pos) else len(data)])
            subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={b}', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                            f'--output={o}'], check=True, capture_output=True)
            txt = subprocess.run([f'{LLVM}/llvm-objdump', '-d', o], check=True, capture_output=True, text=True).stdout
            res.append((f'{os.path.basename(so_path)}#{i}', txt.splitlines()))
    return res


def check_path(path, verbose=False):
    units = device_disassembly(path) if path.endswith('.so') else [(path, open(path).read().splitlines())]
    total, n_tr = {}, 0
    for name, lines in units:
        pk, n = check_lines(lines, name, verbose)
        n_tr += n
        for k, v in pk.items():
            total[k] = total.get(k, 0) + v
    return total, n_tr


def check_path_vmem(path, verbose=False):
    units = device_disassembly(path) if path.endswith('.so') else [(path, open(path).read().splitlines())]
    total, n_ld = {}, 0
    for name, lines in units:
        pk, n = check_vmem_lines(lines, name, verbose)
        n_ld += n
        for k, v in pk.items():
            total[k] = total.get(k, 0) + v
    return total, n_ld


if __name__ == '__main__':
    verbose = '-v' in sys.argv
    bad = 0
    for path in [a for a in sys.argv[1:] if a != '-v']:
        total, n_tr = check_path(path, verbose)
        for k, n in total.items():
            print(f'{n:5d}  {k}')
        bad += sum(total.values())
        print(f'{path}: {n_tr} transposing reads scanned, {sum(total.values())} instructions touch an in-flight destination')
        total, n_ld = check_path_vmem(path, verbose)
        for k, n in total.items():
            print(f'{n:5d}  {k}')
        bad += sum(total.values())
        print(f'{path}: {n_ld} global loads of the attention kernels followed to their wait, {sum(total.values())} instructions touch an in-flight destination')
    sys.exit(1 if bad else 0)
