"""the bench line quotes PMC counters (roofline.traffic, mfma_busy_frac_pmc, hbm_bytes_per_image) only
from a committed rocprofv3 summary whose recorded kernel-source hashes match the sources it runs with.  Round 3 shipped a
line with those keys null because a comment edit after the last profile changed a whole-tree hash.  Now (i) the match is
per file and on comment-/whitespace-stripped code, (ii) only the files the dominant conv kernel is compiled from decide
whether its per-launch counters are quoted, and (iii) this test fails on the CPU box when the newest summary of the bench
configuration does not match HEAD's conv kernel sources -- i.e. tools/round_profiles.sh must be the last thing that
touches them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_code_hash_ignores_comments_and_whitespace():
    import bench
    a = 'int f(int x) {\n  // a comment\n  return x /* inline */ + 1;   asm("s_nop 0 // not a comment");\n}\n'
    b = 'int f(int x){return x+1;asm("s_nop 0 // not a comment");}'
    assert bench._code_only(a) == bench._code_only(b)
    assert bench._code_only(a) != bench._code_only(b.replace('+1', '+2'))
    assert bench._code_only('x = "a  b";') != bench._code_only('x = "a b";')        # string literals are code


def test_newest_summary_of_the_bench_configuration_matches_the_conv_kernel_sources():
    import bench
    ctr = bench.counters_from_profiles('f16x3', 128)       # the default: 2 concurrent sub-batches of 128, one stream profiled
    assert ctr is not None, ('no profiles/*_summary.json of `--ways 1 --batch 128` was taken with the current code of %s: '
                             'run tools/round_profiles.sh on the GPU box and commit profiles/' % (bench.CONV_KERNEL_FILES,))
    assert ctr['traffic'] and ctr['mfma_busy_frac'] and ctr['hbm_bytes_per_image']
    d = json.load(open(os.path.join(ROOT, ctr['file'])))
    now = bench.kernel_source_hashes()
    for f in bench.CONV_KERNEL_FILES:
        assert d['source_hashes'][f] == now[f], f
    # the whole-forward figure sums over every kernel: say which sources moved since (informational, printed by bench.py)
    print('profile %s; sources changed since: %s' % (ctr['file'], ctr['files_changed_since']))
