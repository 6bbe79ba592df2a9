#!/usr/bin/env python3
"""Fixtures of the reference's in-memory API (fermi.h:119-123), made HERE with the reference compiled in place: `fermi example` is the
reference's own caller of fm6_api_unitig / fm6_api_correct (example.c:29-45).
  special.api_l20.mag.gz : fermi example -l 20 special.fq.gz      (fm6_build2 does not trim palindromes: differs from `fermi unitig`)
  tiny.api_ec_k17.fq.gz  : fermi example -eU -k 17 tiny.fq.gz     (fm6_api_correct + fm6_api_writeseq)
(`fermi example -l 50 tiny.fq.gz` is byte for byte tiny.mag.gz -- no palindromes there -- and needs no file of its own.)
Usage: python tests/golden/make_golden_api.py"""
import gzip, os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "fermi")
for out, args in (("special.api_l20.mag.gz", ["-l", "20", "special.fq.gz"]), ("tiny.api_ec_k17.fq.gz", ["-eU", "-k", "17", "tiny.fq.gz"])):
    data = subprocess.run([REF, "example"] + args[:-1] + [os.path.join(HERE, args[-1])], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    with gzip.GzipFile(os.path.join(HERE, out), "wb", mtime=0) as f:
        f.write(data)
    print(out, len(data))
assert subprocess.run([REF, "example", "-l", "50", os.path.join(HERE, "tiny.fq.gz")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout == gzip.open(os.path.join(HERE, "tiny.mag.gz")).read()
