"""Loop statistics of a gfx950 kernel's ISA (hipcc -S output): for every natural loop (a backward branch to a label)
the instruction count by mnemonic.  Used by the experiments in profiles/ and by tools/isa_audit.py.

    python tools/isa_loops.py file.s k_point_scalarmul_ct [--min 500]
"""
import collections
import re
import sys


def is_instr(line):
    s = line.strip()
    return bool(s) and not s.startswith(';') and not s.startswith('.') and not s.endswith(':') and not s.startswith('//')


def functions(text):
    """name -> list of lines, for every symbol that has a body ('name:' ... '.Lfunc_endN:')"""
    lines = text.split('\n')
    out = {}
    cur, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r'^([A-Za-z_][A-Za-z0-9_.$]*):', l)
        if m and not m.group(1).startswith('.L'):
            cur, start = m.group(1), i + 1
        elif cur and re.match(r'^\.Lfunc_end\d+:', l):
            out[cur] = lines[start:i]
            cur = None
    return out


def loops(body):
    """[(first_line, last_line, branch mnemonic, label)] of the backward branches of a function body"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    found = []
    for i, l in enumerate(body):
        m = re.match(r'\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            found.append((labels[m.group(2)], i, m.group(1), m.group(2)))
    return found


def instructions(body, a, b):
    return [l.strip() for l in body[a:b + 1] if is_instr(l)]


def histogram(ins):
    return collections.Counter(l.split()[0] for l in ins)


if __name__ == '__main__':
    text = open(sys.argv[1]).read()
    kernel = sys.argv[2]
    least = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 500
    body = functions(text)[kernel]
    for a, b, br, lab in loops(body):
        ins = instructions(body, a, b)
        if len(ins) < least:
            continue
        h = histogram(ins)
        valu = sum(v for k, v in h.items() if k.startswith('v_'))
        print('%s  lines %d-%d  %s  %d instructions, %d VALU, %d s_nop' % (lab, a, b, br, len(ins), valu, h.get('s_nop', 0)))
        print('   ', ', '.join('%s %d' % kv for kv in h.most_common(24)))
