"""Writes tools/phase_stamps/conv_direct_stamped.hip: conv_direct.hip with a cycle-counter stamp at every phase boundary of every
workgroup (slot 0 kernel entry; per channel slice cc, slots 1 + 5 cc + {0 start, 1 source patch parked, 2 pieces blended,
3 pieces written, 4 product steps done}; slot 31 end) in a device array, and two extern "C" calls to clear and read it.
See build.sh and conv_phases.py."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "monocular-depth-estimation_amd", "csrc", "conv_direct.hip")).read()


def rep(a, b):
    global s
    assert s.count(a) >= 1, a[:60]
    s = s.replace(a, b, 1)


rep("namespace vda {\n\nnamespace {\n", """namespace vda {

constexpr int STAMP_WGS = 40000;
__device__ unsigned long long stamp_buf[STAMP_WGS * 32];
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < STAMP_WGS && blockIdx.y == 0) stamp_buf[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)

namespace {
""")
rep("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n", "    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    STAMP(0);\n")
rep("""        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (also the weights of this step, needed next anyway)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
""", """        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (also the weights of this step, needed next anyway)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(1 + cc * 5 + 1);
""")
rep("""        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                                 // every parked value has been read
""", """        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(1 + cc * 5 + 2);
        __builtin_amdgcn_s_barrier();                                                 // every parked value has been read
""")
rep("""        if (tg == 0) {                                       // all reads of the previous slice ended at the last barrier
""", """        if (tg == 0) {                                       // all reads of the previous slice ended at the last barrier
            STAMP(1 + cc * 5 + 0);
""")
rep("""                issue_halo(cc);
            }
        }
""", """                issue_halo(cc);
            }
            STAMP(1 + cc * 5 + 3);
        }
""")
rep("""        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue""", """        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tg == SPC - 1) STAMP(1 + cc * 5 + 4);
    }

    // ---- epilogue""")
rep("""                reinterpret_cast<float*>(p.out)[((size_t)img * p.out_h + oy) * p.out_w + ox] = fmaxf(part + p.dot_b, 0.f);
        }
        return;""", """                reinterpret_cast<float*>(p.out)[((size_t)img * p.out_h + oy) * p.out_w + ox] = fmaxf(part + p.dot_b, 0.f);
        }
        STAMP(31);
        return;""")
rep("""        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int TN, int TPS, bool RELU_IN, bool UP, int SPEC, typename T, bool PAIRS = false>
int launch_cfg(""", """        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    STAMP(31);
}

template <int TN, int TPS, bool RELU_IN, bool UP, int SPEC, typename T, bool PAIRS = false>
int launch_cfg(""")
i = s.rindex("}  // namespace vda")
s = s[:i] + """extern "C" int vda_stamps_read(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(stamp_buf), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int vda_stamps_clear() {
    void* p;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(stamp_buf)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(stamp_buf));
}
""" + s[i:]
open(os.path.join(ROOT, "tools", "phase_stamps", "conv_direct_stamped.hip"), "w").write(s)
