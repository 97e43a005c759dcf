// Exploration harness (dev tool): HBM streaming structures for out = a + b on 1e8 floats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<bool NTL> __device__ __forceinline__ v4f ld(const v4f* p){ if constexpr(NTL) return __builtin_nontemporal_load(p); else return *p; }
template<bool NTS> __device__ __forceinline__ void st(v4f* p, v4f v){ if constexpr(NTS) __builtin_nontemporal_store(v,p); else *p=v; }

// MODE 0: add (2R+1W), 1: copy (1R+1W), 2: read-only (sum into sink), 3: write-only
// grid-strided, unrolled accesses one grid apart
template<int U, bool NTL, bool NTS, int MODE, int T>
__global__ __launch_bounds__(T) void k_grid(const v4f* a, const v4f* b, v4f* o, unsigned nvec, float* sink){
  unsigned stride = gridDim.x*T; unsigned tid = blockIdx.x*T+threadIdx.x; v4f accs{0,0,0,0};
  for(unsigned base=tid; base<nvec; base+=stride*U){
    v4f va[U], vb[U];
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*stride; if(v<nvec){ if(MODE!=3) va[u]=ld<NTL>(a+v); if(MODE==0) vb[u]=ld<NTL>(b+v);} }
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*stride; if(v<nvec){ if(MODE==0) st<NTS>(o+v, va[u]+vb[u]); else if(MODE==1) st<NTS>(o+v, va[u]); else if(MODE==2) accs+=va[u]; else st<NTS>(o+v, v4f{1,2,3,4}); } }
  }
  if(MODE==2 && accs[0]+accs[1]+accs[2]+accs[3]==12345.678f) sink[0]=1;
}
// block-contiguous chunks: block handles U*T consecutive float4, loops over chunks grid-strided
template<int U, bool NTL, bool NTS, int MODE, int T>
__global__ __launch_bounds__(T) void k_chunk(const v4f* a, const v4f* b, v4f* o, unsigned nvec, float* sink){
  v4f accs{0,0,0,0};
  const unsigned chunk = U*T; unsigned nchunks = (nvec + chunk-1)/chunk;
  for(unsigned c=blockIdx.x; c<nchunks; c+=gridDim.x){
    unsigned base = c*chunk + threadIdx.x;
    v4f va[U], vb[U];
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*T; if(v<nvec){ if(MODE!=3) va[u]=ld<NTL>(a+v); if(MODE==0) vb[u]=ld<NTL>(b+v);} }
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*T; if(v<nvec){ if(MODE==0) st<NTS>(o+v, va[u]+vb[u]); else if(MODE==1) st<NTS>(o+v, va[u]); else if(MODE==2) accs+=va[u]; else st<NTS>(o+v, v4f{1,2,3,4}); } }
  }
  if(MODE==2 && accs[0]+accs[1]+accs[2]+accs[3]==12345.678f) sink[0]=1;
}

// LDS-DMA read side: both operands travel global -> LDS by global_load_lds_dwordx4
// (no destination VGPRs while the loads are in flight), come back with ds_read_b128, the sum leaves by a plain /
// nt store.  Every wave owns its LDS slots (2 x U KiB), so no barrier: s_waitcnt vmcnt(0) orders a wave's own DMA.
template<int U, bool NTS, int AUX, int T>
__global__ __launch_bounds__(T) void k_ldsdma(const v4f* a, const v4f* b, v4f* o, unsigned nvec){
  extern __shared__ float lds[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* wa = lds + wave*(2*U*256); float* wb = wa + U*256;
  const unsigned stride = gridDim.x*T; const unsigned tid = blockIdx.x*T+threadIdx.x;
  for(unsigned base=tid; base<nvec; base+=stride*U){
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*stride; if(v<nvec){
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a+v), (__attribute__((address_space(3))) void*)(wa+u*256), 16, 0, AUX);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b+v), (__attribute__((address_space(3))) void*)(wb+u*256), 16, 0, AUX); } }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*stride; if(v<nvec){
      v4f va = *(const v4f*)(wa+u*256+lane*4), vb = *(const v4f*)(wb+u*256+lane*4);
      st<NTS>(o+v, va+vb); } }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slots are re-filled by the next trip
  }
}

// U accesses per lane, `span` float4 apart (instead of one grid apart): span = ceil(nvec / U) rounded up to a block
// multiple PLUS a stagger, so that a lane's U accesses to one operand do not share their low address bits.
template<int U, int T>
__global__ __launch_bounds__(T) void k_span(const v4f* a, const v4f* b, v4f* o, unsigned nvec, unsigned span){
  const unsigned stride = gridDim.x*T;
  for(unsigned base=blockIdx.x*T+threadIdx.x; base<span; base+=stride){
    v4f va[U], vb[U];
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*span; if(v<nvec){ va[u]=ld<true>(a+v); vb[u]=ld<true>(b+v);} }
    #pragma unroll
    for(int u=0;u<U;++u){ unsigned v=base+u*span; if(v<nvec) st<true>(o+v, va[u]+vb[u]); }
  }
}

float *A,*B,*O,*S; unsigned nvec; hipStream_t st_;
template<typename F> void bench(const char* name, double bytes, F launch){
  for(int i=0;i<3;++i) launch(); CK(hipStreamSynchronize(st_));
  std::vector<float> ts; hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int i=0;i<15;++i){ CK(hipEventRecord(e0,st_)); launch(); CK(hipEventRecord(e1,st_)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ts.push_back(ms);} 
  std::sort(ts.begin(), ts.end());
  // the same kernel as bench.py times it: 25 launches back to back between one event pair (no host sync in between)
  CK(hipEventRecord(e0,st_)); for(int i=0;i<25;++i) launch(); CK(hipEventRecord(e1,st_)); CK(hipEventSynchronize(e1)); float bms; CK(hipEventElapsedTime(&bms,e0,e1)); bms/=25;
  printf("%-44s med %.4f ms  %.0f GB/s   min %.4f ms %.0f GB/s   back-to-back x25 %.4f ms %.0f GB/s\n", name, ts[7], bytes/ts[7]/1e6, ts[0], bytes/ts[0]/1e6, bms, bytes/bms/1e6); fflush(stdout);
}
template<int U,bool NTL,bool NTS,int MODE,int T> void run_cfg(int kind, int bpc){
  double bytes = (MODE==0?12.0:MODE==1?8.0:4.0)*nvec*4;
  unsigned grid;
  if(kind==0){ size_t need=((size_t)nvec+ (size_t)U*T-1)/((size_t)U*T); grid = bpc? std::min<size_t>(need,(size_t)256*bpc):need; }
  else { size_t need=((size_t)nvec+(size_t)U*T-1)/((size_t)U*T); grid = bpc? std::min<size_t>(need,(size_t)256*bpc):need; }
  char name[128]; snprintf(name,128,"%s mode%d U%d T%d ntl%d nts%d bpc%d grid%u", kind?"chunk":"grid ", MODE,U,T,(int)NTL,(int)NTS,bpc,grid);
  if(kind==0) bench(name, bytes, [&]{ k_grid<U,NTL,NTS,MODE,T><<<grid,T,0,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec,S); });
  else bench(name, bytes, [&]{ k_chunk<U,NTL,NTS,MODE,T><<<grid,T,0,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec,S); });
}
template<int U,int T,int MODE> void sweep_nt(int kind,int bpc){ run_cfg<U,true,true,MODE,T>(kind,bpc); run_cfg<U,false,false,MODE,T>(kind,bpc); run_cfg<U,true,false,MODE,T>(kind,bpc); run_cfg<U,false,true,MODE,T>(kind,bpc);} 
__global__ void k_rand(float* p, size_t n, unsigned seed){ size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(; i<n; i+=st){ unsigned long long z = (i + 0x9E3779B97F4A7C15ull*(seed+1)); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; p[i] = (float)(z >> 40) * (1.0f/16777216.0f); } }
template<int U,bool NTL,bool NTS,int T> void run_off(size_t offb, size_t offo){
  double bytes = 12.0*nvec*4; size_t need=((size_t)nvec+(size_t)U*T-1)/((size_t)U*T); unsigned grid=need;
  char name[128]; snprintf(name,128,"offset U%d T%d offB=%zuB offO=%zuB", U,T,offb,offo);
  bench(name, bytes, [&]{ k_grid<U,NTL,NTS,0,T><<<grid,T,0,st_>>>((v4f*)A,(v4f*)((char*)B+offb),(v4f*)((char*)O+offo),nvec,S); });
}
template<int U,bool NTS,int AUX,int T> void run_lds(int bpc){
  double bytes = 12.0*nvec*4; size_t need=((size_t)nvec+(size_t)U*T-1)/((size_t)U*T); unsigned grid = bpc? std::min<size_t>(need,(size_t)256*bpc):need;
  char name[128]; snprintf(name,128,"ldsdma U%d T%d nts%d aux%d bpc%d grid%u", U,T,(int)NTS,AUX,bpc,grid);
  size_t shm = (size_t)(T/64)*2*U*1024;
  bench(name, bytes, [&]{ k_ldsdma<U,NTS,AUX,T><<<grid,T,shm,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec); });
}
template<int U,int T> void run_span(unsigned stagger){
  double bytes = 12.0*nvec*4; unsigned span = ((nvec + U - 1)/U + T - 1)/T*T + stagger; unsigned grid = (span + T - 1)/T;
  char name[128]; snprintf(name,128,"span  U%d T%d stagger %u float4 (%u B)", U,T,stagger,stagger*16);
  bench(name, bytes, [&]{ k_span<U,T><<<grid,T,0,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec,span); });
}
__global__ void k_check(const float* a, const float* b, const float* o, size_t n, unsigned* bad){ size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(; i<n; i+=st) if(o[i]!=a[i]+b[i]) atomicAdd(bad,1u); }
int main(int argc, char** argv){ if(argc>1 && argv[1][0]=='s'){
  // ./add_bw slab : the three operands carved out of ONE hipMalloc at chosen distances (is the run-to-run spread of
  // separately allocated buffers physical placement?  does a fixed relative layout remove it?)
  size_t n=100000000; nvec=n/4; CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  const size_t MiB = 1u<<20; char* slab; CK(hipMalloc(&slab, (size_t)1600*MiB)); CK(hipMalloc(&S,4));
  printf("slab=%p\n", slab);
  const size_t base_sp = 384*MiB;   // 400,000,000 B rounded up to 2 MiB
  size_t spacings[] = {base_sp, base_sp+1024, base_sp+4096, base_sp+65536, base_sp+2*MiB, base_sp+2*MiB+1024, base_sp+8*MiB, base_sp+8*MiB+1024, base_sp+32*MiB+1024, 400*MiB+1024, 448*MiB+1024, 512*MiB, 512*MiB+1024};
  for(int rep=0; rep<2; ++rep) for(size_t sp: spacings){
    for(int order=0; order<2; ++order){
      A=(float*)(slab + (order? 2*sp:0)); B=(float*)(slab+sp); O=(float*)(slab + (order? 0:2*sp));
      if(rep==0 && order==0){ k_rand<<<2048,256,0,st_>>>((float*)slab,(size_t)1600*MiB/4,1); CK(hipStreamSynchronize(st_)); }
      char name[128]; snprintf(name,128,"slab spacing %4zu MiB + %6zu B  %s", sp/MiB, sp%MiB, order? "o<b<a":"a<b<o");
      double bytes = 12.0*nvec*4; size_t need=((size_t)nvec+511)/512; unsigned grid=need|1;
      bench(name, bytes, [&]{ k_grid<2,true,true,0,256><<<grid,256,0,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec,S); });
    }
  }
  return 0; }
  if(argc>1 && argv[1][0]=='l'){
  // ./add_bw lds : LDS-DMA read side vs the library's structure (grid U2 T256 nt/nt, uncapped), random data
  size_t n=100000000; nvec=n/4; CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  CK(hipMalloc(&A,n*4)); CK(hipMalloc(&B,n*4)); CK(hipMalloc(&O,n*4)); CK(hipMalloc(&S,4));
  printf("A=%p B=%p O=%p (hipMalloc: 2 MiB aligned = %d %d %d)\n",A,B,O,(int)(((size_t)A&0x1fffff)==0),(int)(((size_t)B&0x1fffff)==0),(int)(((size_t)O&0x1fffff)==0));
  k_rand<<<2048,256,0,st_>>>(A,n,1); k_rand<<<2048,256,0,st_>>>(B,n,2); CK(hipStreamSynchronize(st_));
  // correctness of the DMA kernel first
  CK(hipMemsetAsync(O,0,n*4,st_)); CK(hipMemsetAsync(S,0,4,st_));
  { const int U=2,T=256; size_t need=((size_t)nvec+(size_t)U*T-1)/((size_t)U*T); k_ldsdma<U,true,0,T><<<need,T,(T/64)*2*U*1024,st_>>>((v4f*)A,(v4f*)B,(v4f*)O,nvec); }
  k_check<<<4096,256,0,st_>>>(A,B,O,n,(unsigned*)S); unsigned bad=1; CK(hipMemcpyAsync(&bad,S,4,hipMemcpyDeviceToHost,st_)); CK(hipStreamSynchronize(st_));
  printf("ldsdma U2 correctness: %u mismatches of %zu\n", bad, n);
  for(int rep=0; rep<3; ++rep){
    printf("-- round %d\n", rep);
    run_cfg<2,true,true,0,256>(0,0);            // the library's add structure
    run_cfg<4,true,true,1,256>(0,0); run_cfg<4,true,true,2,256>(0,0); run_cfg<4,true,true,3,256>(0,0);   // copy / read / write ceilings
    run_lds<1,true,0,256>(0); run_lds<2,true,0,256>(0); run_lds<4,true,0,256>(0); run_lds<8,true,0,256>(0);
    run_lds<2,true,2,256>(0); run_lds<4,true,2,256>(0); run_lds<2,false,0,256>(0); run_lds<2,true,0,512>(0); run_lds<4,true,0,128>(0);
    run_lds<4,true,0,256>(8); run_lds<4,true,0,256>(16); run_lds<8,true,0,256>(4);
    // second sweep (the nt hint on the DMA was the only variant ahead of the plain kernel in the first)
    run_lds<1,true,2,256>(0); run_lds<3,true,2,256>(0); run_lds<6,true,2,256>(0); run_lds<4,true,2,512>(0); run_lds<4,true,2,128>(0);
    run_lds<4,true,3,256>(0); run_lds<4,true,18,256>(0); run_lds<4,true,19,256>(0); run_lds<4,false,2,256>(0);
    run_cfg<4,true,true,0,256>(0,0); run_cfg<1,true,true,0,256>(0,0);
    // third sweep: U accesses per lane `span` apart with a stagger between them
    run_span<2,256>(0); run_span<2,256>(16); run_span<2,256>(64); run_span<2,256>(128); run_span<2,256>(192); run_span<2,256>(1024+64);
    run_span<4,256>(0); run_span<4,256>(64); run_span<3,256>(64); run_span<1,256>(0);
  }
  return 0; }
  if(argc>2){
  size_t n=100000000; nvec=n/4; CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  CK(hipMalloc(&A,n*4+(8<<20))); CK(hipMalloc(&B,n*4+(8<<20))); CK(hipMalloc(&O,n*4+(8<<20))); CK(hipMalloc(&S,4));
  printf("A=%p B=%p O=%p\n",A,B,O);
  k_rand<<<2048,256,0,st_>>>(A,n,1); k_rand<<<2048,256,0,st_>>>(B,n+1000000,2); CK(hipStreamSynchronize(st_));
  for(int rep=0; rep<2; ++rep){
    for(size_t off: {(size_t)0,(size_t)256,(size_t)1024,(size_t)4096,(size_t)16384,(size_t)65536,(size_t)262144,(size_t)1048576,(size_t)(1048576+4096+256)}) run_off<2,true,true,256>(off, 2*off);
  }
  return 0; }
  if(argc>1){
  size_t n=100000000; nvec=n/4; CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  CK(hipMalloc(&A,n*4)); CK(hipMalloc(&B,n*4)); CK(hipMalloc(&O,n*4)); CK(hipMalloc(&S,4));
  k_rand<<<2048,256,0,st_>>>(A,n,1); k_rand<<<2048,256,0,st_>>>(B,n,2); CK(hipStreamSynchronize(st_));
  for(int rep=0; rep<3; ++rep){
    run_cfg<4,true,true,0,64>(0,0); run_cfg<4,true,true,0,256>(0,0); run_cfg<1,true,true,0,256>(1,0); run_cfg<4,true,true,0,128>(0,0);
    run_cfg<8,true,true,0,64>(0,0); run_cfg<2,true,true,0,64>(0,0); run_cfg<2,true,true,0,128>(0,0); run_cfg<2,true,true,0,256>(0,0); run_cfg<1,true,true,0,64>(0,0);
    run_cfg<4,true,true,0,1024>(0,0); run_cfg<1,true,true,0,1024>(0,0); run_cfg<4,true,true,1,256>(0,0); run_cfg<4,true,true,2,256>(0,0);
  }
  return 0; }
  size_t n=100000000; nvec=n/4; CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  CK(hipMalloc(&A,n*4)); CK(hipMalloc(&B,n*4)); CK(hipMalloc(&O,n*4)); CK(hipMalloc(&S,4));
  CK(hipMemsetAsync(A,0x3f,n*4,st_)); CK(hipMemsetAsync(B,0x3e,n*4,st_)); CK(hipMemsetAsync(O,0,n*4,st_));
  // reference rooflines on this box
  for(int bpc: {0,2,4,8,16}) { run_cfg<4,true,true,1,256>(1,bpc); run_cfg<4,true,true,2,256>(1,bpc); run_cfg<4,true,true,3,256>(1,bpc);} 
  // add: structure x bpc
  for(int kind: {0,1}) for(int bpc: {0,1,2,3,4,6,8,16}) { sweep_nt<4,256,0>(kind,bpc); }
  for(int kind: {0,1}) for(int bpc: {0,1,2,4,8}) { run_cfg<1,true,true,0,256>(kind,bpc); run_cfg<2,true,true,0,256>(kind,bpc); run_cfg<8,true,true,0,256>(kind,bpc); run_cfg<16,true,true,0,256>(kind,bpc);} 
  for(int kind: {0,1}) for(int bpc: {0,1,2,4}) { run_cfg<2,true,true,0,512>(kind,bpc); run_cfg<4,true,true,0,512>(kind,bpc); run_cfg<2,true,true,0,1024>(kind,bpc); run_cfg<4,true,true,0,1024>(kind,bpc); run_cfg<4,true,true,0,64>(kind,bpc*4); run_cfg<8,true,true,0,128>(kind,bpc*2);} 
  return 0;
}
