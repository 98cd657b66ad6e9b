#!/usr/bin/env python3
"""Resolve the lab-only conditionals of the kernel sources: `#ifdef AVT_LAB` blocks (cycle stamps, environment switches, negative-result kernel
variants) and `#if <A/B switch>` branches are evaluated with the product's values and the dead side removed; every other conditional
(__HIP_DEVICE_COMPILE__ ...) is left alone.  Used once in round 6 to take those blocks out of avt_amd/csrc (the removed text lives on as
tools/lab/avt_lab_hooks.diff); kept as the tool that did it.   usage: tools/strip_lab.py file ... (in place)"""
import re
import sys

UNDEF = {'AVT_LAB', 'AVT_OLD_SMALL_TILE_RULE', 'AVT_ATTN_BWD_TWO_PHASE', 'AVT_PK_STAGGER'}
VALUES = {'PK_FULLPF': 0, 'AVT_PK_ABL': 0, 'AVT_ATTN_WIDE_ST': 1, 'AVT_ATTN_ABL': 0, 'AVT_SLAB_NT': 0}


def evaluate(kind, expr):
    """True / False when the condition is decided by the tables above, None = leave the directive alone."""
    expr = expr.split('//')[0].strip()
    if kind in ('ifdef', 'ifndef'):
        if expr in UNDEF:
            return kind == 'ifndef'
        return None
    m = re.fullmatch(r'defined\s*\(\s*(\w+)\s*\)', expr)
    if m and m.group(1) in UNDEF:
        return False
    names = set(re.findall(r'[A-Za-z_]\w*', expr))
    if names and names <= (set(VALUES) | UNDEF):
        e = expr
        for n in names:
            e = re.sub(r'\b%s\b' % n, str(VALUES.get(n, 0)), e)
        return bool(eval(e, {'__builtins__': {}}))          # integer expressions only: & | > < == digits
    return None


def strip(text):
    out, stack = [], []          # stack entries: [decided (bool | None), emitting_now (bool), any_branch_taken (bool)]
    for line in text.split('\n'):
        m = re.match(r'\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)', line)
        live = all(s[1] for s in stack)
        if not m:
            if live:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ('ifdef', 'ifndef', 'if'):
            v = evaluate(kind, rest) if live else False
            if not live:
                stack.append(['dead', False, True])
            elif v is None:
                stack.append([None, True, False]); out.append(line)
            else:
                stack.append([True, bool(v), bool(v)])
        elif kind in ('elif', 'else'):
            top = stack[-1]
            if top[0] == 'dead':
                continue
            if top[0] is None:
                if all(s[1] for s in stack[:-1]):
                    out.append(line)
                continue
            if kind == 'else':
                top[1] = not top[2]; top[2] = True
            else:
                v = evaluate('if', rest)
                assert v is not None, 'undecidable #elif inside a decided #if: ' + line
                top[1] = (not top[2]) and bool(v); top[2] = top[2] or top[1]
        else:
            top = stack.pop()
            if top[0] is None and all(s[1] for s in stack):
                out.append(line)
    assert not stack, 'unbalanced conditionals'
    return '\n'.join(out)


if __name__ == '__main__':
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        if new != src:
            open(path, 'w').write(new)
            print(f'{path}: {src.count(chr(10)) - new.count(chr(10))} lines removed')
