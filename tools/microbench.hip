// tools/microbench.hip — diagnostic only: launch/ingest floors on MI355X for the NFP launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/microbench tools/microbench.hip && ./gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && out == nullptr) out[0] = 1; }

struct Big { int a[40]; long long b[6]; };
// does a ~220-byte by-value kernarg block cost anything?  fields are consumed in dependent order
__global__ void k_bigargs(Big k, float* out) {
  int s = k.a[0];
  s = k.a[(s & 1) + 17] + s;              // dependent scalar loads
  s += (int)k.b[(s & 3) + 1];
  s += k.a[(s & 7) + 30];
  if (s == 123456 && threadIdx.x == 0) out[blockIdx.x] = (float)s;
}
// LDS round trip + a few barriers, no global traffic at all
__global__ void k_barriers(float* out, int nbar) {
  extern __shared__ float4 lds[];
  float acc = 0;
  for (int i = 0; i < nbar; ++i) {
    ((float*)lds)[threadIdx.x] = acc + i;
    __syncthreads();
    acc += ((float*)lds)[(threadIdx.x * 7 + i) % blockDim.x];
    __syncthreads();
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}
// one dependent global load chain of length n per thread (latency probe under an idle chip)
__global__ void k_chain(const int* __restrict__ idx, float* out, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  for (int k = 0; k < n; ++k) i = idx[i];
  if (i == -7) out[0] = 1;
}
// every thread pulls `n16` 16-byte pieces of its workgroup's contiguous slab, sums, one store per thread
template <int N16>
__global__ void k_ingest(const float4* __restrict__ x, float* __restrict__ out, int slab16) {
  const float4* s = x + (size_t)blockIdx.x * slab16;
  float4 v[N16];
#pragma unroll
  for (int i = 0; i < N16; ++i) v[i] = s[min((int)(threadIdx.x + i * blockDim.x), slab16 - 1)];
  float a = 0;
#pragma unroll
  for (int i = 0; i < N16; ++i) a += v[i].x + v[i].y + v[i].z + v[i].w;
  if (a == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// the NFP NCHW staging pattern: item i -> (cq = i / 12, pq = i % 12); 4 loads of 16 B at 4-byte-aligned
// addresses (c*49 + 4*pq), c = 4cq+j; plus the tail pixel (48) with 4 dword loads
template <int RB>
__global__ void k_ingest_nchw(const float* __restrict__ x, float* __restrict__ out, int ncq) {
  const float* s = x + (size_t)blockIdx.x * ncq * 4 * 49;
  float4 v[RB][4];
  float tl[4];
  const int nblk = ncq * 12;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int i = min((int)(threadIdx.x + r * blockDim.x), nblk - 1);
    int cq = i / 12, pq = i - cq * 12;
    const float* e = s + (size_t)(4 * cq) * 49 + 4 * pq;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[r][j] = *(const float4*)(e + j * 49);
  }
  {
    int cq = min((int)threadIdx.x, ncq - 1);
    const float* e = s + (size_t)(4 * cq) * 49 + 48;
#pragma unroll
    for (int j = 0; j < 4; ++j) tl[j] = e[j * 49];
  }
  float a = tl[0] + tl[1] + tl[2] + tl[3];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) a += v[r][j].x + v[r][j].y + v[r][j].z + v[r][j].w;
  if (a == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// p-consecutive dword loads: thread (p = t % 49, g = t / 49), channels 4(g + kG)+j
template <int K>
__global__ void k_ingest_dword(const float* __restrict__ x, float* __restrict__ out, int ncq, int G) {
  const float* s = x + (size_t)blockIdx.x * ncq * 4 * 49;
  const int p = threadIdx.x % 49, g = min((int)(threadIdx.x / 49), G - 1);
  float a = 0;
  float v[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    int cq = min(g + k * G, ncq - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[k][j] = s[(size_t)(4 * cq + j) * 49 + p];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) a += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  if (a == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// ingest + write back (copy through registers)
template <int N16>
__global__ void k_copy(const float4* __restrict__ x, float4* __restrict__ y, int slab16) {
  const float4* s = x + (size_t)blockIdx.x * slab16;
  float4* d = y + (size_t)blockIdx.x * slab16;
  float4 v[N16];
#pragma unroll
  for (int i = 0; i < N16; ++i) v[i] = s[min((int)(threadIdx.x + i * blockDim.x), slab16 - 1)];
#pragma unroll
  for (int i = 0; i < N16; ++i) if (threadIdx.x + i * blockDim.x < slab16) d[threadIdx.x + i * blockDim.x] = v[i];
}
// ingest -> LDS -> barrier -> read back neighbours (9 b128 per slot) -> store
template <int N16>
__global__ void k_stencil(const float4* __restrict__ x, float4* __restrict__ y, int slab16) {
  extern __shared__ float4 lds[];
  const float4* s = x + (size_t)blockIdx.x * slab16;
  float4* d = y + (size_t)blockIdx.x * slab16;
  float4 v[N16];
#pragma unroll
  for (int i = 0; i < N16; ++i) v[i] = s[min((int)(threadIdx.x + i * blockDim.x), slab16 - 1)];
#pragma unroll
  for (int i = 0; i < N16; ++i) if (threadIdx.x + i * blockDim.x < slab16) lds[threadIdx.x + i * blockDim.x] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N16; ++i) {
    int e = threadIdx.x + i * blockDim.x;
    if (e < slab16) {
      float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        float4 q = lds[min(e + j, slab16 - 1)];
        r.x += q.x * (j + 1); r.y += q.y * (j + 1); r.z += q.z * (j + 1); r.w += q.w * (j + 1);
      }
      d[e] = r;
    }
  }
}

template <typename F>
float time_us(F launch, hipStream_t st, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < reps; ++i) launch();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int k = 0; k < 5; ++k) {
    hipEventRecord(a, st); hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best * 1e3f / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t bytes = 64ull * 512 * 49 * 4;  // 6.4 MB
  float4 *x, *y; float* o;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&o, 1 << 22));
  CK(hipMemset(x, 0, bytes));
  const int reps = 50;
  printf("empty grid64x1024: %.2f us\n", time_us([&] { k_empty<<<64, 1024, 0, st>>>(o); }, st, reps));
  printf("empty grid256x512: %.2f us\n", time_us([&] { k_empty<<<256, 512, 0, st>>>(o); }, st, reps));
  printf("empty grid256x512 lds50K: %.2f us\n", time_us([&] { k_empty<<<256, 512, 50 * 1024, st>>>(o); }, st, reps));
  printf("empty grid64x1024 lds100K: %.2f us\n", time_us([&] { hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024); k_empty<<<64, 1024, 100 * 1024, st>>>(o); }, st, reps));
  printf("empty grid1024x256: %.2f us\n", time_us([&] { k_empty<<<1024, 256, 0, st>>>(o); }, st, reps));
  {
    Big big; for (int i = 0; i < 40; ++i) big.a[i] = i; for (int i = 0; i < 6; ++i) big.b[i] = i;
    printf("bigargs grid256x512: %.2f us\n", time_us([&] { k_bigargs<<<256, 512, 0, st>>>(big, o); }, st, reps));
    for (int nb : {1, 4, 16}) {
      printf("barriers x%d grid256x512: %.2f us\n", 2 * nb, time_us([&] { k_barriers<<<256, 512, 4096, st>>>(o, nb); }, st, reps));
      printf("barriers x%d grid64x1024: %.2f us\n", 2 * nb, time_us([&] { k_barriers<<<64, 1024, 8192, st>>>(o, nb); }, st, reps));
    }
    int* idx; hipMalloc(&idx, 256 * 512 * 4);
    std::vector<int> h(256 * 512); for (int i = 0; i < 256 * 512; ++i) h[i] = (i * 97 + 13) % (256 * 512);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int n : {1, 2, 4, 8})
      printf("dependent load chain x%d grid256x512: %.2f us\n", n, time_us([&] { k_chain<<<256, 512, 0, st>>>(idx, o, n); }, st, reps));
  }
  // ingest 6.4 MB with different shapes
  auto ing = [&](int wgs, int thr, auto kern, int n16) {
    int slab16 = (int)(bytes / 16 / wgs);
    float t = time_us([&] { kern<<<wgs, thr, 0, st>>>(x, o, slab16); }, st, reps);
    printf("ingest  %4d WG x %4d thr x %2d x16B : %.2f us  (%.2f TB/s)\n", wgs, thr, n16, t, bytes / t * 1e-6);
  };
  ing(64, 1024, k_ingest<7>, 7);
  ing(128, 1024, k_ingest<4>, 4);
  ing(256, 1024, k_ingest<2>, 2);
  ing(256, 512, k_ingest<4>, 4);
  ing(512, 512, k_ingest<2>, 2);
  ing(1024, 256, k_ingest<2>, 2);
  ing(2048, 256, k_ingest<1>, 1);
  printf("nchw-pattern ingest 64 WG x 1024 (128 quads, 2 rounds): %.2f us\n", time_us([&] { k_ingest_nchw<2><<<64, 1024, 0, st>>>((const float*)x, o, 128); }, st, reps));
  printf("nchw-pattern ingest 256 WG x 512 (32 quads, 1 round): %.2f us\n", time_us([&] { k_ingest_nchw<1><<<256, 512, 0, st>>>((const float*)x, o, 32); }, st, reps));
  printf("dword-pattern ingest 64 WG x 1024 (G=20,K=7): %.2f us\n", time_us([&] { k_ingest_dword<7><<<64, 1024, 0, st>>>((const float*)x, o, 128, 20); }, st, reps));
  printf("dword-pattern ingest 256 WG x 512 (G=10,K=4): %.2f us\n", time_us([&] { k_ingest_dword<4><<<256, 512, 0, st>>>((const float*)x, o, 32, 10); }, st, reps));
  auto cp = [&](int wgs, int thr, auto kern, int n16) {
    int slab16 = (int)(bytes / 16 / wgs);
    float t = time_us([&] { kern<<<wgs, thr, 0, st>>>(x, y, slab16); }, st, reps);
    printf("copy    %4d WG x %4d thr x %2d x16B : %.2f us  (%.2f TB/s r+w)\n", wgs, thr, n16, t, 2 * bytes / t * 1e-6);
  };
  cp(256, 512, k_copy<4>, 4);
  cp(512, 512, k_copy<2>, 2);
  cp(1024, 256, k_copy<2>, 2);
  cp(2048, 256, k_copy<1>, 1);
  auto stn = [&](int wgs, int thr, auto kern, int n16) {
    int slab16 = (int)(bytes / 16 / wgs);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    float t = time_us([&] { kern<<<wgs, thr, slab16 * 16, st>>>(x, y, slab16); }, st, reps);
    printf("stencil %4d WG x %4d thr x %2d x16B : %.2f us\n", wgs, thr, n16, t);
  };
  stn(256, 512, k_stencil<4>, 4);
  stn(512, 512, k_stencil<2>, 2);
  stn(512, 256, k_stencil<4>, 4);
  stn(1024, 256, k_stencil<2>, 2);
  stn(2048, 256, k_stencil<1>, 1);
  return 0;
}
