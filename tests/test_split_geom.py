"""The wide loads of the split-read comparison's reverse-complement branch (conga_amd/csrc/split_geom.h; the reference compares base
by base, split_read.c:158-203) never begin in front of the reference text: every (c, n) on the host, no GPU.

Round 3's soak met a GPU memory fault there on its last day (seed 82, case 11: a half of more than 56 bases whose reverse complement
maps within 56 bases of chromosome 1's first base; commit 4000c64).  The guard and the load's arithmetic are two functions of a
header that compiles for the host; this test fails on the guard as it was (`c + n >= 64`) and passes on the one the kernel calls."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r'''
#include "%s/conga_amd/csrc/split_geom.h"
#include <cstdio>
static bool old_guard(int c, int n) { return c + n >= 56 + 8; }
int main()
{
	long wide = 0, bad_new = 0, bad_old = 0, first_old_c = -1, first_old_n = -1;
	for (int n = 10; n <= 1022; n++)          // (a half-read buffer holds up to 1 022 bases; the seed is ten)
		for (int c = 0; c <= 300; c++) {
			int lowest = 1 << 30;
			for (int k0 = 0; k0 < n; k0 += 56)
				if (sr_rev_step_base(c, n, k0) < lowest)
					lowest = sr_rev_step_base(c, n, k0);
			if (sr_rev_wide_ok(c, n)) {
				wide++;
				if (lowest < 0)
					bad_new++;
			}
			if (old_guard(c, n) && lowest < 0) {
				if (!bad_old) { first_old_c = c; first_old_n = n; }
				bad_old++;
			}
		}
	printf("wide %%ld bad_new %%ld bad_old %%ld first_old %%ld %%ld\n", wide, bad_new, bad_old, first_old_c, first_old_n);
	return 0;
}
'''


def test_no_wide_load_begins_in_front_of_the_reference_text(tmp_path):
    src = tmp_path / "g.cpp"
    src.write_text(HARNESS % ROOT)
    exe = tmp_path / "g"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    vals = dict(zip(out[0::2][:3], map(int, out[1::2][:3])))
    assert vals["wide"] > 200_000        # the wide form is the rule (only windows at a chromosome's very start go base by base)
    assert vals["bad_new"] == 0          # the guard the kernel calls: no load begins in front of base 0
    assert vals["bad_old"] > 1000        # round 3's guard let such loads through -- the first one at c = 7 with a half of 57 bases (c < 7 went base by base)
    assert out[-2:] == ["7", "57"]


def test_the_kernel_calls_the_tested_functions():
    text = open(os.path.join(ROOT, "conga_amd", "csrc", "split_map.hip.h")).read()
    assert "sr_rev_wide_ok(c, n) ? half_distance_rev(" in text and "const int t = sr_rev_step_base(c, n, k0);" in text
