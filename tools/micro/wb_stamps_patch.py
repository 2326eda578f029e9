"""Diagnostic build of the window-attention backward (winattn_bwd_dma) with s_memtime stamps per wave and phase.

    python tools/micro/wb_stamps_patch.py apply     # patches grit_amd/csrc/winattn.hip IN PLACE (keeps winattn.hip.orig) and rebuilds
    python tools/micro/wb_stamps.py                 # on the GPU box: cycles per workgroup, wave and phase -> profiles/r03/winattn_bwd_stamps.txt
    python tools/micro/wb_stamps_patch.py restore   # puts the product source back and rebuilds

The stamps add a global symbol and ~40 scalar instructions per window: never commit the patched source.  The anchors are source
lines of the kernel; the script asserts each of them, so it fails loudly when the kernel has moved on."""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "grit_amd", "csrc", "winattn.hip")


def apply():
    shutil.copy(SRC, SRC + ".orig")
    p = SRC
    s=open(p).read()
    i=s.index('template <bool kExplicitMask>\n__global__ __launch_bounds__(kThreads)\nvoid winattn_bwd_dma(')
    head, body = s[:i], s[i:]
    st = {'body': body}
    head+="__device__ unsigned long long g_wb_stamps[256];\n#define WB_ST(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - tprev; tprev = now_; }\n"
    def rep(a, b):
        assert a in st['body'], a
        st['body'] = st['body'].replace(a, b, 1)
    rep("    Next nxt;\n    int cur = 0;\n","    Next nxt;\n    int cur = 0;\n    unsigned long long st[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long tprev = __builtin_amdgcn_s_memtime();\n")
    rep("        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // this wave's DMAs (and its stores of the previous window) are done\n        __syncthreads();",
      "        WB_ST(0)\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // this wave's DMAs (and its stores of the previous window) are done\n        WB_ST(1)\n        __syncthreads();\n        WB_ST(2)")
    rep("        __syncthreads();  // statistics visible; the V / O tiles and the other tile buffer are free for the next window's DMA\n",
      "        WB_ST(3)\n        __syncthreads();  // statistics visible; the V / O tiles and the other tile buffer are free for the next window's DMA\n        WB_ST(4)\n")
    rep("        // Per-lane LDS offsets, made opaque once per window","        WB_ST(5)\n        // Per-lane LDS offsets, made opaque once per window")
    rep("        if (tkk >= 0) {\n            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;\n            v4bf a, c, e, f;",
      "        WB_ST(6)\n        if (tkk >= 0) {\n            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;\n            v4bf a, c, e, f;")
    rep("        __syncthreads();  // dSt complete\n","        WB_ST(7)\n        __syncthreads();  // dSt complete\n        WB_ST(8)\n")
    rep("    // ---- flush the register-resident d(bias) of this workgroup's windows and the padded-token gradient",
      "    if (lane == 0) { for (int k = 0; k < 10; ++k) atomicAdd(&g_wb_stamps[w * 16 + k], st[k]); atomicAdd(&g_wb_stamps[w * 16 + 10], 1ull); }\n    // ---- flush the register-resident d(bias) of this workgroup's windows and the padded-token gradient")
    rep("                if (l15 == 0) {\n                    atomicAdd(&pad_s[4 * lg + r], a);\n                    atomicAdd(&pad_s[16 + 4 * lg + r], c);\n                }\n            }\n        }\n    }\n",
      "                if (l15 == 0) {\n                    atomicAdd(&pad_s[4 * lg + r], a);\n                    atomicAdd(&pad_s[16 + 4 * lg + r], c);\n                }\n            }\n        }\n        WB_ST(9)\n    }\n")
    s = head + st['body']
    s+="\nextern \"C\" int grit_debug_winattn_stamps(unsigned long long* out, int reset) {\n    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wb_stamps), sizeof(unsigned long long) * 256) != hipSuccess) return -1;\n    if (reset) { unsigned long long z[256] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wb_stamps), z, sizeof(z)) != hipSuccess) return -1; }\n    return 0;\n}\n"
    open(p,'w').write(s)


def restore():
    shutil.move(SRC + ".orig", SRC)


if __name__ == "__main__":
    {"apply": apply, "restore": restore}[sys.argv[1]]()
    from grit_amd import build
    build.build_all()
