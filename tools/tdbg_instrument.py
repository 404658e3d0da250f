#!/usr/bin/env python3
"""Builds a diagnostic copy of the library whose chain launch records, per workgroup, start / end time (wall_clock64),
HW_ID and XCC_ID (i.e. which compute unit it ran on), and writes them as CSV to the path in $CONGA_TDBG after every
compute.  Usage (scratch tree, nothing in the repo is modified):

    python tools/tdbg_instrument.py /tmp/tdbg && CONGA_TDBG=/tmp/wg.csv CONGA_LIB_PATH=/tmp/tdbg/libconga_tdbg.so \\
        python bench.py --steps 2 --warmup 1 --results-on-device --cpu-seconds 0 --no-dense-leg
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sub(text, old, new):
    assert old in text, old[:60]
    return text.replace(old, new, 1)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/tdbg"
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "conga_amd"))
    shutil.copytree(os.path.join(ROOT, "conga_amd", "csrc"), os.path.join(out, "conga_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(out, "include"))
    kp = os.path.join(out, "conga_amd", "csrc", "kernels.hip.h")
    k = open(kp).read()
    k = sub(k, "	int32_t zero_blocks;    // > 0:", "	unsigned long long *tdbg;\n	int32_t zero_blocks;    // > 0:")
    stamp = ("a.tdbg[4 * blockIdx.x] = t_start; a.tdbg[4 * blockIdx.x + 1] = wall_clock64(); "
             "a.tdbg[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(63492); "
             "a.tdbg[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(63508);")
    k = sub(k, "	int b = (int) blockIdx.x;\n	if (b < (int) a.n_x) {",
            "	int b = (int) blockIdx.x;\n	const unsigned long long t_start = wall_clock64();\n	if (b < (int) a.n_x) {")
    k = sub(k, "		chain_block_body(a, (int64_t) b, 0, a.n_x, sE, xw);\n		return;",
            "		chain_block_body(a, (int64_t) b, 0, a.n_x, sE, xw);\n		if (a.tdbg && threadIdx.x == 0) { %s }\n		return;" % stamp)
    k = sub(k, "a.n_iv - a.n_x - a.n_a - a.n_b, sE, stage);\n	}\n",
            "a.n_iv - a.n_x - a.n_a - a.n_b, sE, stage);\n	}\n	if (a.tdbg) {\n		__syncthreads();\n		if (threadIdx.x == 0) { %s }\n	}\n" % stamp)
    open(kp, "w").write(k)
    ap = os.path.join(out, "conga_amd", "csrc", "conga_api.hip")
    a = open(ap).read()
    launch = ("			hipLaunchKernelGGL(interval_chain_kernel, dim3((int) c.n_x + c.blocks_ab + blocks_c + c.zero_blocks + "
              "c.table_blocks), dim3(256), 0,\n					st, c);")
    a = sub(a, launch, r'''			static unsigned long long *tdbg_dev = nullptr;
			c.tdbg = nullptr;
			if (getenv("CONGA_TDBG")) {
				if (!tdbg_dev)
					(void) hipMalloc((void **) &tdbg_dev, 16384 * 32);
				c.tdbg = tdbg_dev;
			}
''' + launch + r'''
			if (c.tdbg) {
				const int n_work = (int) c.n_x + c.blocks_ab + blocks_c;
				std::vector<unsigned long long> t(4 * (size_t) n_work);
				(void) hipStreamSynchronize(st);
				(void) hipMemcpy(t.data(), c.tdbg, t.size() * 8, hipMemcpyDeviceToHost);
				unsigned long long t0 = ~0ull;
				for (int b2 = 0; b2 < n_work; b2++)
					t0 = std::min(t0, t[4 * b2]);
				if (FILE *f = fopen(getenv("CONGA_TDBG"), "w")) {
					fprintf(f, "wg,class,start_us,end_us,hwid,xcc\n");
					for (int b2 = 0; b2 < n_work; b2++) {
						const char *cls = b2 < (int) c.n_x ? "X" : b2 < (int) c.n_x + (int) std::min<int64_t>(c.n_a, c.blocks_ab) ? "AB"
								: b2 < (int) c.n_x + c.blocks_ab ? "B" : "C";
						fprintf(f, "%d,%s,%.2f,%.2f,%llu,%llu\n", b2, cls, (t[4 * b2] - t0) / 100.0, (t[4 * b2 + 1] - t0) / 100.0,
								t[4 * b2 + 2], t[4 * b2 + 3]);
					}
					fclose(f);
				}
			}''')
    open(ap, "w").write(a)
    so = os.path.join(out, "libconga_tdbg.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fno-gpu-flush-denormals-to-zero", "-o", so, ap])
    print(so)


if __name__ == "__main__":
    main()
