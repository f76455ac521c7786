// valu_rates.hip -- issue cost of the VALU instructions the render kernels are made of, on the machine at hand.
// Every kernel runs dependent chains (4 independent chains per lane to cover the pipeline latency) of one instruction,
// one workgroup of 256 threads per CU (one wave per SIMD), and reports cycles per wave-instruction from s_memtime-free
// wall_clock64 (100 MHz) against the launch's instruction count.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 4096;

// four independent dependent-chains per lane, 8 x 4 instructions per loop iteration, written as inline assembly so that
// the compiler neither folds nor reorders them
#define CHAIN_KERNEL(name, T, C, INSTR)                                                                   \
    __global__ void __launch_bounds__(256) name(T* out, T seed) {                                         \
        T a = seed + (T)threadIdx.x, b = a + (T)1, c = a + (T)2, d = a + (T)3;                            \
        for (int i = 0; i < kIters; ++i) {                                                                \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                 \
                asm volatile(INSTR(0, 4) "\n" INSTR(1, 4) "\n" INSTR(2, 4) "\n" INSTR(3, 4)               \
                             : "+" C(a), "+" C(b), "+" C(c), "+" C(d) : C(seed));                         \
        }                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                              \
    }
#define REG_V(x) "v"(x)
#define I2(op) op " %" 
#define OP3(op, k, s) op " %" #k ", %" #k ", %" #s
#define ADD_F64(k, s) OP3("v_add_f64", k, s)
#define MUL_F64(k, s) OP3("v_mul_f64", k, s)
#define FMA_F64(k, s) "v_fma_f64 %" #k ", %" #k ", %" #s ", %" #s
#define ADD_F32(k, s) OP3("v_add_f32", k, s)
#define FMA_F32(k, s) "v_fma_f32 %" #k ", %" #k ", %" #s ", %" #s
#define RSQ_F64(k, s) "v_rsq_f64 %" #k ", %" #k
#define RCP_F64(k, s) "v_rcp_f64 %" #k ", %" #k
#define MIN_F64(k, s) OP3("v_min_f64", k, s)
#define CMP_F64(k, s) "v_cmp_lt_f64 vcc, %" #k ", %" #s "\n v_cndmask_b32 %" #k ", %" #k ", %" #s ", vcc"
#define MULLO(k, s) OP3("v_mul_lo_u32", k, s)
#define MULHI(k, s) OP3("v_mul_hi_u32", k, s)
#define MAD64(k, s) "v_mad_u64_u32 %" #k ", vcc, %" #s ", %" #s ", %" #k
#define XOR32(k, s) OP3("v_xor_b32", k, s)
#define ADD32(k, s) OP3("v_add_u32", k, s)
#define CVT_F64_F32(k, s) "v_cvt_f32_f64 %" #s ", %" #k

CHAIN_KERNEL(k_add_f64, double, REG_V, ADD_F64)
CHAIN_KERNEL(k_mul_f64, double, REG_V, MUL_F64)
CHAIN_KERNEL(k_fma_f64, double, REG_V, FMA_F64)
CHAIN_KERNEL(k_add_f32, float, REG_V, ADD_F32)
CHAIN_KERNEL(k_fma_f32, float, REG_V, FMA_F32)
CHAIN_KERNEL(k_rsq_f64, double, REG_V, RSQ_F64)
CHAIN_KERNEL(k_rcp_f64, double, REG_V, RCP_F64)
CHAIN_KERNEL(k_min_f64, double, REG_V, MIN_F64)
CHAIN_KERNEL(k_mullo_u32, unsigned, REG_V, MULLO)
CHAIN_KERNEL(k_mulhi_u32, unsigned, REG_V, MULHI)
CHAIN_KERNEL(k_xor_u32, unsigned, REG_V, XOR32)
CHAIN_KERNEL(k_add_u32, unsigned, REG_V, ADD32)

// v_mad_u64_u32: 64-bit accumulator, 32-bit factors
__global__ void __launch_bounds__(256) k_mad_u64(unsigned long long* out, unsigned long long seed) {
    unsigned long long a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const unsigned f = (unsigned)seed | 1u;
    for (int i = 0; i < kIters; ++i) {
        _Pragma("unroll") for (int u = 0; u < 8; ++u)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\nv_mad_u64_u32 %1, vcc, %4, %4, %1\nv_mad_u64_u32 %2, vcc, %4, %4, %2\nv_mad_u64_u32 %3, vcc, %4, %4, %3"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(f) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

template <typename T, typename K>
void run(const char* name, K kernel, T seed, int per_iter, int waves_per_simd) {
    int dev = 0, cus = 0, mhz = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev);  // kHz
    T* out = nullptr;
    const int blocks = cus * waves_per_simd;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(out, seed);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        kernel<<<blocks, 256>>>(out, seed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // per SIMD: waves_per_simd waves, each kIters * 8 * per_iter instructions of interest
    const double instr = (double)kIters * 8 * per_iter * waves_per_simd;
    const double cycles = best * 1e-3 * (double)mhz * 1e3;
    std::printf("%-14s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction (at %d MHz)\n", name, waves_per_simd, best, cycles / instr, mhz / 1000);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<double>("v_add_f64", k_add_f64, 1.000001, 4, w);
        run<double>("v_mul_f64", k_mul_f64, 1.000001, 4, w);
        run<double>("v_fma_f64", k_fma_f64, 1.000001, 4, w);
        run<float>("v_add_f32", k_add_f32, 1.000001f, 4, w);
        run<float>("v_fma_f32", k_fma_f32, 1.000001f, 4, w);
        run<double>("v_rsq_f64", k_rsq_f64, 1.5, 4, w);
        run<double>("v_rcp_f64", k_rcp_f64, 1.5, 4, w);
        run<double>("v_min_f64", k_min_f64, 1.5, 4, w);
        run<unsigned>("v_mul_lo_u32", k_mullo_u32, 0xCD9E8D57u, 4, w);
        run<unsigned>("v_mul_hi_u32", k_mulhi_u32, 0xCD9E8D57u, 4, w);
        run<unsigned long long>("v_mad_u64_u32", k_mad_u64, 12345ull, 4, w);
        run<unsigned>("v_xor_b32", k_xor_u32, 0xCD9E8D57u, 4, w);
        run<unsigned>("v_add_u32", k_add_u32, 0xCD9E8D57u, 4, w);
    }
    return 0;
}
