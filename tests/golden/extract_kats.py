#!/usr/bin/env python3
"""Re-type the reference's own known-answer vectors as data (runs in the build container only).

Source (read, never copied as text): /root/reference/test/vectors.inc.cxx (RFC 8032 Ed448 x11,
RFC 7748 X448 iterated) and test/elligator_vectors.inc.cxx (k*B decaf encodings, k = 0..15).
Output: tests/golden/kats.json -- byte strings as hex, nothing else.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def byte_arrays(text):
    """name -> list of byte strings, for `const uint8_t name[...]... = { ... };` definitions."""
    out = {}
    for m in re.finditer(r"const\s+uint8_t\s+([\w:<>]+?)\s*((?:\[[^\]]*\])+)\s*=\s*\{(.*?)\};", text, re.S):
        name, dims, body = m.group(1).split("::")[-1], m.group(2), m.group(3)
        if dims.count("[") == 2:
            rows = re.findall(r"\{([^{}]*)\}", body)
        else:
            rows = [body]
        out[name] = [bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{1,2})", r)) for r in rows]
    return out


def block_tables(text):
    """name -> list of (array, index, length) / None, for `...::name[] = { Block(a[i],n), ... };`"""
    out = {}
    for m in re.finditer(r"const\s+Block\s+[\w:<>]+::(\w+)\[\]\s*=\s*\{(.*?)\};", text, re.S):
        items = []
        for b in re.finditer(r"Block\(\s*(NULL|(\w+)\[(\d+)\])\s*,\s*(\d+)\s*\)", m.group(2)):
            items.append(None if b.group(1) == "NULL" else (b.group(2), int(b.group(3)), int(b.group(4))))
        out[m.group(1)] = items
    return out


def main():
    v = strip_comments(open(os.path.join(REF, "test", "vectors.inc.cxx")).read())
    e = strip_comments(open(os.path.join(REF, "test", "elligator_vectors.inc.cxx")).read())
    arrs, blocks = byte_arrays(v), block_tables(v)
    pre = re.search(r"eddsa_prehashed\[\]\s*=\s*\{(.*?)\};", v, re.S).group(1)
    prehashed = [w == "true" for w in re.findall(r"true|false", pre)]

    def get(tab, i):
        ref = blocks[tab][i]
        if ref is None:
            return b""
        name, idx, ln = ref
        return arrs[name][idx][:ln]

    cases = []
    for i in range(len(prehashed)):
        cases.append({"sk": get("eddsa_sk", i).hex(), "pk": get("eddsa_pk", i).hex(),
                      "message": get("eddsa_message", i).hex(), "context": get("eddsa_context", i).hex(),
                      "prehashed": prehashed[i], "sig": get("eddsa_sig", i).hex()})
    assert len(cases) == 11 and all(len(c["sig"]) == 228 and len(c["pk"]) == 114 for c in cases)
    ea = byte_arrays(e)
    bm = ea["values"]
    assert len(bm) == 16 and all(len(b) == 56 for b in bm)
    ell_in, ell_out = ea["inputs"], ea["outputs"]
    assert len(ell_in) == 16 and len(ell_out) == 16 and all(len(b) == 56 for b in ell_in + ell_out)
    kats = {
        "source": "otrv4/libgoldilocks test/vectors.inc.cxx:3-751, test/elligator_vectors.inc.cxx:3-217",
        "rfc8032_ed448": cases,
        "base_multiples": [b.hex() for b in bm],
        "elligator_nonuniform": [{"hash": i.hex(), "point": o.hex()} for i, o in zip(ell_in, ell_out)],
        "rfc7748_x448_iterated": {"1": arrs["rfc7748_1"][0].hex(), "1000": arrs["rfc7748_1000"][0].hex(),
                                  "1000000": arrs["rfc7748_1000000"][0].hex()},
    }
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote kats.json: %d Ed448 cases, %d base multiples" % (len(cases), len(bm)))


if __name__ == "__main__":
    main()
