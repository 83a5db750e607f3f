#!/usr/bin/env python3
"""Writes fma_banks.hip: issue rate of v_fma_f64 / v_mul_f64 on gfx950 as a function of WHICH vector registers the
operands live in (register-file banks), with 1, 2, 4 and 8 waves per SIMD.  Straight-line assembly, fixed registers."""
from pathlib import Path

N_INSTR = 256  # per loop iteration


def block(kind: str) -> list[str]:
    """8 independent accumulators / 8 sources; bank of a 64-bit pair = (first register mod 4) in {0, 2}."""
    out = []
    for i in range(N_INSTR):
        j = i % 8
        if kind == "same":      # acc and source pairs both start at a multiple of 4
            acc, src = 8 + 4 * j, 40 + 4 * ((j + 3) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], s[4:5], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        elif kind == "diff":    # acc pairs start at 2 mod 4, sources at 0 mod 4
            acc, src = 10 + 4 * j, 40 + 4 * ((j + 3) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], s[4:5], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        elif kind == "vvv_same":  # three vector operands, all pairs at 0 mod 4
            acc, src, c = 8 + 4 * j, 40 + 4 * ((j + 3) % 8), 40 + 4 * ((j + 5) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], v[{c}:{c+1}], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        elif kind == "vvv_diff":  # acc at 2 mod 4, sources 0 mod 4 and 2 mod 4
            acc, src, c = 10 + 4 * j, 40 + 4 * ((j + 3) % 8), 42 + 4 * ((j + 5) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], v[{c}:{c+1}], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        elif kind == "mul":
            acc, src = 10 + 4 * j, 40 + 4 * ((j + 3) % 8)
            out.append(f"v_mul_f64 v[{acc}:{acc+1}], s[4:5], v[{src}:{src+1}]")
        elif kind == "inplace":  # a = s * a + b, b in the other bank (the butterfly's last operations)
            acc, src = 8 + 4 * j, 42 + 4 * ((j + 3) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], s[4:5], v[{acc}:{acc+1}], v[{src}:{src+1}]")
        elif kind == "f32":
            acc, src = 8 + j, 40 + ((j + 3) % 8)
            out.append(f"v_fma_f32 v{acc}, s4, v{src}, v{acc}")
        elif kind == "mov64":
            acc, src = 8 + 4 * j, 42 + 4 * ((j + 3) % 8)
            out.append(f"v_mov_b64 v[{acc}:{acc+1}], v[{src}:{src+1}]")
        elif kind == "dpp":
            acc, src = 8 + j, 40 + ((j + 3) % 8)
            out.append(f"v_mov_b32_dpp v{acc}, v{src} row_shr:4 row_mask:0xf bank_mask:0xc")
        elif kind == "permlane":
            acc, src = 8 + j, 40 + ((j + 3) % 8)
            out.append(f"v_permlane32_swap_b32 v{acc}, v{src}")
        elif kind == "dep4":   # every operation depends on the one four slots earlier (as in the butterfly)
            acc, src = 10 + 4 * (i % 4), 40 + 4 * ((j + 3) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], s[4:5], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        elif kind == "dep2":
            acc, src = 10 + 4 * (i % 2), 40 + 4 * ((j + 3) % 8)
            out.append(f"v_fma_f64 v[{acc}:{acc+1}], s[4:5], v[{src}:{src+1}], v[{acc}:{acc+1}]")
        else:
            raise ValueError(kind)
    return out


KINDS = ["same", "diff", "vvv_same", "vvv_diff", "mul", "inplace", "f32", "mov64", "dpp", "permlane", "dep4", "dep2"]


def main() -> None:
    src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>', '']
    clob = ", ".join(f'"v{i}"' for i in range(8, 76)) + ', "s4", "s5", "s6", "scc"'
    for k in KINDS:
        src.append(f"__global__ void __launch_bounds__(256) k_{k}(double* out, int iters, double c) {{")
        src.append("    asm volatile(")
        src.append('        "s_mov_b64 s[4:5], %[c]\\n\\t"')
        src.append('        "s_mov_b32 s6, %[n]\\n\\t"')
        for r in range(8, 76):
            src.append(f'        "v_mov_b32 v{r}, 0\\n\\t"')
        src.append('        "Lloop_%=:\\n\\t"')
        for line in block(k):
            src.append(f'        "{line}\\n\\t"')
        src.append('        "s_sub_u32 s6, s6, 1\\n\\t"')
        src.append('        "s_cmp_lg_u32 s6, 0\\n\\t"')
        src.append('        "s_cbranch_scc1 Lloop_%=\\n\\t"')
        src.append(f'        : : [c] "s"(c), [n] "s"(iters) : {clob});')
        src.append("    if (iters < 0) out[threadIdx.x] = c;")
        src.append("}")
        src.append("")
    src.append("typedef void (*kern_t)(double*, int, double);")
    src.append("struct K { const char* name; kern_t f; };")
    src.append("static const K kinds[] = {" + ", ".join(f'{{"{k}", k_{k}}}' for k in KINDS) + "};")
    src.append(f"static const int N_INSTR = {N_INSTR};")
    src.append(r'''
int main(int argc, char** argv) {
    double* d; (void)hipMalloc(&d, 4096);
    const int iters = 2000;
    for (const K& k : kinds) {
        if (argc > 1 && strcmp(argv[1], k.name)) continue;
        printf("%-10s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;  // 4 waves per workgroup: one per SIMD
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 10, 0.5);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = double(wps) * iters * N_INSTR;  // wave-instructions each SIMD issued
            printf("  %d w/SIMD: %5.2f cyc/instr", wps, ms * 1e-3 * 2.4e9 / per_simd);
        }
        printf("\n");
    }
    return 0;
}
''')
    Path(__file__).with_name("fma_banks.hip").write_text("\n".join(src))


if __name__ == "__main__":
    main()
