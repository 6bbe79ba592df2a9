"""GPU, BASELINE.json configs[4] (700 M x 100 bp, human-35x scale: 1.4*10^11 symbols, one GPU's 1/8 share of the ids) and the
builders that only sizes beyond 2^32 symbols exercise, inside `pytest -m gpu` (they were builder-run tool logs).

  * 7*10^8 reads: the in-place builder (4-bit text + BWT slices ORed straight into the device layout, prefix buckets of depth 4),
    the rank self-check over all 1.4*10^11 positions, sampled reads hit themselves, overlap discovery of random ids checked by what
    the generator knows (sequences, the one neighbour and its overlap from the start positions, mutual edges, mirrored intervals),
    the four-part k-mer harvest cross-checked by backward search, the share i = 0 (mod 8) of the discovery timed -- and `unitig` END TO END: the .fmd written, `fermi-amd
    unitig -l50` on it as its own process, the MAG checked exactly against the generator (189 unitigs, 2.33*10^9 bases; 80 s, 65 GB resident).  ~8 minutes.
  * 1.3*10^8 reads WITH 1 % substitutions (2.6*10^10 symbols): the bucketed byte-BWT builder, the .fmd written by the product and loaded by the
    REFERENCE (oracle/_ref when it travelled, the oracle otherwise): backward search and overlap discovery of 20 000 random reads / ids
    bit-exact, the sorted job on 200 000 random ids (forks: the general group kernels and the 64-bit fast kernels beyond 2^32 symbols) bit-exact,
    check_left against the oracle.  ~4 minutes.
Both run tools/scale_check.py (the log goes to gpurun_out/ when that directory exists)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scale_check(tag, args, timeout):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_check.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    txt = p.stdout.decode(errors="replace")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "pytest_scale_%s.txt" % tag), "w").write(txt)
    assert p.returncode == 0 and "scale check passed" in txt, txt[-3000:]
    return txt


@pytest.mark.skipif(os.environ.get("FMD_TEST_SKIP_CFG5") == "1", reason="FMD_TEST_SKIP_CFG5=1")
def test_config5_700m_reads_index_in_place_and_one_gpus_share(gpu):
    import torch
    free_b, total_b = torch.cuda.mem_get_info()
    assert total_b > 250e9, "config 5 needs the 288 GB of an MI355X"
    txt = _scale_check("700M", ["700000000", "inplace", "20000", "8", "noref", "fmd", "kmer", "props", "dry", "unitig", "genmag"], 1700)
    # the 8-rank step's buffers at this size, walked on one rank beside the index (FMD_DIST_DRY): the root's table in pinned host memory, a peer's staging sets
    # (root and a peer, key shard and id shard; pieces as small as they have to be beside 153 GB of index -- the sizes are in the log, gpurun_out/pytest_scale_700M.txt)
    assert txt.count("every allocation succeeded") == 4 and txt.count("every buffer of a step at full size") == 4, txt[-3000:]
    assert "141400000000 positions: 0 bad" in txt and "properties on" in txt and "cross-checked by backward search" in txt
    assert "share 1/8 of the overlap discovery on this index: 175000000 strands" in txt and "(0 overflow records" in txt
    # `unitig` of config 5 END TO END (missing 1): the .fmd of the in-place builder written (BWT decoded from the device layout, the host's RLD encoder:
    # 24.6 GB), `fermi-amd unitig -l50` on it as its own process (index load, 1.4*10^9 rows through the slim table, link pass, the walk, 4.7 GB of MAG), and the MAG
    # checked EXACTLY against the generator: every unitig's sequence, coverage string and number of reads, no other records (tools/mag_vs_generator.py; the rule is
    # pinned against the reference's own MAGs in tests/test_oracle_golden.py and profiles/r6_cfg5/rule_vs_reference_1M.txt)
    import re
    m = re.search(r"unitig -l50 by fermi-amd: rc 0, ([\d.]+) s, MAG (\d+) bytes md5 \w+; peak resident set ([\d.]+) GB", txt)
    assert m, txt[-3000:]
    assert float(m.group(3)) < 66.0, m.group(0)                       # 62.3 GB of table (44.5 bytes per row) + the runtime + the longest unitig's strings
    assert "table of 1400000000 sequences on 1 GPU(s)" in txt and "0 edges left to the exact kernel" in txt
    assert re.search(r"MAG: (\d+) records, 23333\d+ bases.*; \1 runs expected, \1 matched exactly", txt), txt[-3000:]
    assert "against the generator (every unitig's sequence, coverage string and number of reads; no other records): EXACT" in txt


def test_130m_raw_reads_bucketed_builder_fmd_and_reference(gpu):
    """2.6*10^10 symbols of reads WITH 1 % substitutions: beyond 2^32 symbols the forked path of fm6_get_nei (the general group kernels, the 64-bit fast kernels) meets the
    reference -- config 5 and the 7*10^8-read case above are error-free, where no strand forks."""
    txt = _scale_check("130M_raw", ["130000000", "bwt", "20000", "8", "raw"], 900)
    assert "26260000000 positions: 0 bad" in txt
    assert "backward search vs" in txt and "overlap discovery vs" in txt and "the sorted job vs the reference" in txt and "check_left_simple of" in txt and "MISMATCH" not in txt


def test_130m_reads_unitig_holds_a_third_of_the_packed_table(gpu):
    """`fermi-amd unitig -l50` on 1.3*10^8 error-free reads (2.6*10^8 rows) as its own process: the table the walk runs over is kept slim (host/slim_table.c:
    44.5 bytes per row; rounds 1-4 held the packed rows, 150-190), so the peak resident set of the whole command -- table, staging buffers, bitmaps, the runtime;
    VmHWM of the process, which unlike ru_maxrss does not inherit the test's own -- stays below 15 GB where the earlier CLI measured 50.9 GB on the same .fmd
    (profiles/r5_slim), and the MAG is the one that CLI printed (its md5 is in the log when build/old/fermi-amd-old travelled with the tree)."""
    import re
    txt = _scale_check("130M_unitig", ["130000000", "bwt", "2000", "8", "unitig", "genmag"], 900)
    m = re.search(r"unitig -l50 by fermi-amd: rc 0, [\d.]+ s, MAG (\d+) bytes md5 (\w+); peak resident set ([\d.]+) GB", txt)
    assert m, txt[-3000:]
    # the MAG against the GENERATOR, exactly (tools/mag_vs_generator.py: sequence, coverage string and number of reads of every unitig, no other records) -- until
    # round 6 this line compared the md5 with an earlier CLI's own output; the size is what that MAG had
    assert int(m.group(1)) == 866669808
    assert "against the generator (every unitig's sequence, coverage string and number of reads; no other records): EXACT" in txt, txt[-3000:]
    assert float(m.group(3)) < 15.0, m.group(0)
    assert "bytes per row in host memory" in txt
