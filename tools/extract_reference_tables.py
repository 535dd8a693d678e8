#!/usr/bin/env python3
"""Extract the numeric DATA tables the hot path needs from the reference mount.

Run in the build container only (needs /root/reference and gcc). Outputs are
plain binary data files, committed under rayfinder_amd/data/:

  blue_noise_128x128_rg8.bin  32768 bytes: 128x128 texels x (R,G), top-left origin
                              (reference: src/pt/blue_noise.c, blue_noise.h:11-15)
  hw_sky_tables.bin           little-endian f32, concatenated in this order:
                                params_r[1080] params_g[1080] params_b[1080]
                                radiances_r[120] radiances_g[120] radiances_b[120]
                                solar_radiances_r[10] _g[10] _b[10]
                              (reference: src/hw-skymodel/params_{r,g,b}.h,
                               radiances_{r,g,b}.h - Hosek-Wilkie RGB coefficient data)

No reference source text is copied: a throw-away C program #includes the tables where they
lie under /root/reference and fwrite()s the arrays.
"""
import os, subprocess, sys, tempfile, hashlib

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rayfinder_amd", "data")

PROG = r'''
#include <stdio.h>
#include <stdint.h>
#include <stddef.h>
#include "hw-skymodel/params_r.h"
#include "hw-skymodel/params_g.h"
#include "hw-skymodel/params_b.h"
#include "hw-skymodel/radiances_r.h"
#include "hw-skymodel/radiances_g.h"
#include "hw-skymodel/radiances_b.h"
#include "pt/blue_noise.h"
#define DUMP(f, a) do { fwrite((a), sizeof((a)[0]), sizeof(a)/sizeof((a)[0]), f); \
    fprintf(stderr, "%s %zu\n", #a, sizeof(a)/sizeof((a)[0])); } while (0)
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "wb");
    DUMP(f, params_r); DUMP(f, params_g); DUMP(f, params_b);
    DUMP(f, radiances_r); DUMP(f, radiances_g); DUMP(f, radiances_b);
    DUMP(f, solar_radiances_r); DUMP(f, solar_radiances_g); DUMP(f, solar_radiances_b);
    fclose(f);
    f = fopen(argv[2], "wb");
    fwrite(blueNoiseValues, 1, 32768, f);
    fprintf(stderr, "blue noise %zu x %zu\n", blueNoiseWidth, blueNoiseHeight);
    fclose(f);
    return 0;
}
'''

def main():
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "dump.c")
        open(src, "w").write(PROG)
        exe = os.path.join(td, "dump")
        subprocess.check_call(["gcc", "-O0", "-I", REF, src, os.path.join(REF, "pt/blue_noise.c"), "-o", exe])
        sky = os.path.join(OUT, "hw_sky_tables.bin")
        bn = os.path.join(OUT, "blue_noise_128x128_rg8.bin")
        subprocess.check_call([exe, sky, bn])
    for p in (sky, bn):
        d = open(p, "rb").read()
        print(os.path.basename(p), len(d), hashlib.sha256(d).hexdigest())

if __name__ == "__main__":
    sys.exit(main())
