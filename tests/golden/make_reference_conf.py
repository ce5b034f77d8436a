#!/usr/bin/env python3
"""Extracts the key/value pairs of the reference's two PWN configuration files (g2o_frontend/pwn_core/conf/pwn_aligner_1_{1,4}.conf: the
inputs of pwn_simple_aligner.cpp:190-269) into tests/golden/reference_conf.json, with the parser semantics of pwn_simple_aligner.cpp:190-212:
`key value` per line, lines whose first two tokens do not parse as (word, float) are skipped, the first occurrence of a key wins."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/g2o_frontend/pwn_core/conf"


def parse(path):
    out = {}
    for line in open(path):
        tok = line.split()
        if len(tok) < 2:
            continue
        try:
            v = float(tok[1])
        except ValueError:
            continue
        out.setdefault(tok[0], v)
    return out


if __name__ == "__main__":
    d = {name: parse(os.path.join(REF, name)) for name in ("pwn_aligner_1_1.conf", "pwn_aligner_1_4.conf")}
    json.dump(d, open(os.path.join(HERE, "reference_conf.json"), "w"), indent=1, sort_keys=True)
    print({k: len(v) for k, v in d.items()})
