"""Exhaustive proof-by-enumeration of the kernel's integer tail (CPU, ~3 s with gcc -O2 -fopenmp):
  * requantisation: for every n in [-32768, 32767] and q in [1, 65535], (int)((float)n * rq) with
    rq = smallest float >= 1/q equals C's n / q (what `short /= UINT16` computes, reference
    compose.c:327-336);  2^32 cases, same float arithmetic as libmodjpeg_amd/csrc (one f32 multiply)
  * dequantisation: low 16 bits of the packed 16-bit product equal (short)(coef * q)
    (reference compose.c:277-286) for all coef, sampled q."""
import os
import subprocess
import textwrap

SRC = textwrap.dedent(r'''
    #include <math.h>
    #include <stdint.h>
    #include <stdio.h>
    static float rq_of(unsigned q) { double r = 1.0 / (double)q; float f = (float)r; if((double)f < r) f = nextafterf(f, INFINITY); return f; }
    int main(void) {
        long bad = 0, badmul = 0;
        #pragma omp parallel for reduction(+:bad,badmul) schedule(dynamic, 256)
        for(int q = 1; q <= 65535; q++) {
            const float rq = rq_of((unsigned)q);
            for(int n = -32768; n <= 32767; n++) {
                volatile float p = (float)n * rq;      /* one rounding, no contraction */
                if((int)p != n / q) bad++;
            }
            if(q % 97 == 1 || q > 65000) {
                for(int c = -32768; c <= 32767; c++) {
                    uint16_t pk = (uint16_t)((uint16_t)c * (uint16_t)q);   /* v_pk_mul_lo_u16 lane */
                    int16_t ref = (int16_t)(c * q);                        /* short *= UINT16 */
                    if((int16_t)pk != ref) badmul++;
                }
            }
        }
        printf("%ld %ld\n", bad, badmul);
        return 0;
    }
''')


def test_requant_and_dequant_exhaustive(tmp_path):
    src = tmp_path / "rq.c"
    src.write_text(SRC)
    exe = tmp_path / "rq"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    out = subprocess.check_output([str(exe)], text=True, env=dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1))))
    bad, badmul = [int(v) for v in out.split()]
    assert bad == 0 and badmul == 0
