"""Extracts the known answer of the reference's test_pairing_result_against_relic
(coset-bls12_381/src/tests/mod.rs:66-183: e(G1 generator, G2 generator) as an Fp12 of raw Montgomery limbs) into
pairing_generators.json.  Run once in a checkout that has the reference at /root/reference:
    python tests/golden/make_pairing_vector.py
Order of the 12 Fp values: c0.c0.c0, c0.c0.c1, c0.c1.c0, c0.c1.c1, c0.c2.c0, c0.c2.c1, then the same for c1."""
import json
import os
import re

SRC = "/root/reference/coset-bls12_381/src/tests/mod.rs"
lines = open(SRC).read().split("\n")[65:183]
words = [int(w.replace("_", ""), 16) for w in re.findall(r"0x[0-9a-f_]+", "\n".join(lines))]
assert len(words) == 72, len(words)
fps = [words[6 * i:6 * i + 6] for i in range(12)]
out = {"source": "coset-bls12_381/src/tests/mod.rs:66-183", "form": "six little-endian u64 limbs of a * 2^384 mod p",
       "fp12": [[hex(w) for w in fp] for fp in fps]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pairing_generators.json"), "w"), indent=1)
print("wrote", len(fps), "field elements")
