// Hardware-semantics probe for gfx950: MFMA operand/accumulator layouts, ds_read_tr16_b64,
// LDS-DMA (global_load_lds / buffer_load ... lds) ordering and out-of-bounds behaviour.
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run on the GPU box, prints a report.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

static unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); unsigned r = u + 0x7fff + ((u>>16)&1); return (unsigned short)(r>>16);} 
static float bf2f(unsigned short h){ unsigned u = ((unsigned)h)<<16; float f; memcpy(&f,&u,4); return f; }

__global__ void k_mfma16(const unsigned short* a, const unsigned short* b, float* c) {
  int l = threadIdx.x; bf8 av, bv;
  for (int i=0;i<8;i++){ av[i] = __builtin_bit_cast(__bf16, a[l*8+i]); bv[i] = __builtin_bit_cast(__bf16, b[l*8+i]); }
  f32x4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0,0,0);
  for (int i=0;i<4;i++) c[l*4+i]=acc[i];
}
__global__ void k_mfma32(const unsigned short* a, const unsigned short* b, float* c) {
  int l = threadIdx.x; bf8 av, bv;
  for (int i=0;i<8;i++){ av[i] = __builtin_bit_cast(__bf16, a[l*8+i]); bv[i] = __builtin_bit_cast(__bf16, b[l*8+i]); }
  f32x16 acc; for(int i=0;i<16;i++) acc[i]=0;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0,0,0);
  for (int i=0;i<16;i++) c[l*16+i]=acc[i];
}
__global__ void k_mfma16k16(const unsigned short* a, const unsigned short* b, float* c) {
  int l = threadIdx.x; s16x4 av, bv;
  for (int i=0;i<4;i++){ av[i] = a[l*4+i]; bv[i] = b[l*4+i]; }
  f32x4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc, 0,0,0);
  for (int i=0;i<4;i++) c[l*4+i]=acc[i];
}
// tr read: mode 0: lane-linear addresses; mode 1: [4][16]-blocks inside a row-stride-32 tile
__global__ void k_tr(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  int l = threadIdx.x;
  for (int i=l;i<2048;i+=64) lds[i]=(short)i;
  __syncthreads();
  int off;
  if (mode==0) off = l*4;
  else { int g=l>>4, i=l&15; off = (i>>2)*32 + g*16*0 + (i&3)*4 + g*128; }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int i=0;i<4;i++) out[l*4+i]=v[i];
}
// LDS-DMA via global_load_lds 16B; lane i source = g + perm(i)*8 shorts
__global__ void k_glds(const short* g, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for (int i=l;i<1024;i+=64) lds[i]=-1;
  __syncthreads();
  int src = (63-l);   // reversed source, linear destination
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src*8), (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i=0;i<8;i++) out[l*8+i]=lds[l*8+i];
}
// buffer_load ... lds with bounds: lanes whose offset >= nbytes should produce zeros (or leave LDS untouched?)
__global__ void k_buflds(const short* g, short* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for (int i=l;i<1024;i+=64) lds[i]=-1;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, l*16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i=0;i<8;i++) out[l*8+i]=lds[l*8+i];
}
// buffer load to registers with OOB
__global__ void k_bufreg(const short* g, int* out, int nbytes) {
  int l = threadIdx.x;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, l*16, 0, 0);
  for (int i=0;i<4;i++) out[l*4+i]=v[i];
}

int main(){
  srand(1);
  // ---- MFMA 16x16x32 ----
  {
    std::vector<float> A(16*32), B(32*16); // A[i][k], B[k][j]
    for (auto& x: A) x = bf2f(f2bf((rand()%17-8)/4.f));
    for (auto& x: B) x = bf2f(f2bf((rand()%13-6)/4.f));
    std::vector<unsigned short> a(64*8), b(64*8);
    for (int l=0;l<64;l++) for(int e=0;e<8;e++){ int k=(l>>4)*8+e; a[l*8+e]=f2bf(A[(l&15)*32+k]); b[l*8+e]=f2bf(B[k*16+(l&15)]); }
    unsigned short *da,*db; float* dc; CK(hipMalloc(&da,1024)); CK(hipMalloc(&db,1024)); CK(hipMalloc(&dc,1024));
    CK(hipMemcpy(da,a.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(db,b.data(),1024,hipMemcpyHostToDevice));
    k_mfma16<<<1,64>>>(da,db,dc); CK(hipDeviceSynchronize());
    std::vector<float> c(256); CK(hipMemcpy(c.data(),dc,1024,hipMemcpyDeviceToHost));
    int bad=0; for (int l=0;l<64;l++) for(int r=0;r<4;r++){ int row=(l>>4)*4+r, col=l&15; float ref=0; for(int k=0;k<32;k++) ref+=A[row*32+k]*B[k*16+col]; if (fabs(ref-c[l*4+r])>1e-3) bad++; }
    printf("MFMA16x16x32 hypothesis (A[i=l&15][k=8*(l>>4)+e], B[k][j=l&15], C row=4*(l>>4)+r col=l&15): %s (bad=%d)\n", bad?"FAIL":"OK", bad);
  }
  // ---- MFMA 32x32x16 ----
  {
    std::vector<float> A(32*16), B(16*32);
    for (auto& x: A) x = bf2f(f2bf((rand()%17-8)/4.f));
    for (auto& x: B) x = bf2f(f2bf((rand()%13-6)/4.f));
    std::vector<unsigned short> a(64*8), b(64*8);
    for (int l=0;l<64;l++) for(int e=0;e<8;e++){ int k=(l>>5)*8+e; a[l*8+e]=f2bf(A[(l&31)*16+k]); b[l*8+e]=f2bf(B[k*32+(l&31)]); }
    unsigned short *da,*db; float* dc; CK(hipMalloc(&da,1024)); CK(hipMalloc(&db,1024)); CK(hipMalloc(&dc,4096));
    CK(hipMemcpy(da,a.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(db,b.data(),1024,hipMemcpyHostToDevice));
    k_mfma32<<<1,64>>>(da,db,dc); CK(hipDeviceSynchronize());
    std::vector<float> c(1024); CK(hipMemcpy(c.data(),dc,4096,hipMemcpyDeviceToHost));
    int bad=0; for (int l=0;l<64;l++) for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31; float ref=0; for(int k=0;k<16;k++) ref+=A[row*16+k]*B[k*32+col]; if (fabs(ref-c[l*16+r])>1e-3) bad++; }
    printf("MFMA32x32x16 hypothesis (A[i=l&31][k=8*(l>>5)+e], C row=(r&3)+8*(r>>2)+4*(l>>5) col=l&31): %s (bad=%d)\n", bad?"FAIL":"OK", bad);
  }
  // ---- MFMA 16x16x16 (1k) ----
  {
    std::vector<float> A(16*16), B(16*16);
    for (auto& x: A) x = bf2f(f2bf((rand()%17-8)/4.f));
    for (auto& x: B) x = bf2f(f2bf((rand()%13-6)/4.f));
    std::vector<unsigned short> a(64*4), b(64*4);
    for (int l=0;l<64;l++) for(int e=0;e<4;e++){ int k=(l>>4)*4+e; a[l*4+e]=f2bf(A[(l&15)*16+k]); b[l*4+e]=f2bf(B[k*16+(l&15)]); }
    unsigned short *da,*db; float* dc; CK(hipMalloc(&da,512)); CK(hipMalloc(&db,512)); CK(hipMalloc(&dc,1024));
    CK(hipMemcpy(da,a.data(),512,hipMemcpyHostToDevice)); CK(hipMemcpy(db,b.data(),512,hipMemcpyHostToDevice));
    k_mfma16k16<<<1,64>>>(da,db,dc); CK(hipDeviceSynchronize());
    std::vector<float> c(256); CK(hipMemcpy(c.data(),dc,1024,hipMemcpyDeviceToHost));
    int bad=0; for (int l=0;l<64;l++) for(int r=0;r<4;r++){ int row=(l>>4)*4+r, col=l&15; float ref=0; for(int k=0;k<16;k++) ref+=A[row*16+k]*B[k*16+col]; if (fabs(ref-c[l*4+r])>1e-3) bad++; }
    printf("MFMA16x16x16_1k hypothesis (k=4*(l>>4)+e): %s (bad=%d)\n", bad?"FAIL":"OK", bad);
  }
  // ---- tr read ----
  for (int mode=0; mode<2; mode++) {
    short* dout; CK(hipMalloc(&dout,512)); k_tr<<<1,64>>>(dout,mode); CK(hipDeviceSynchronize());
    std::vector<short> o(256); CK(hipMemcpy(o.data(),dout,512,hipMemcpyDeviceToHost));
    int bad=0;
    for (int l=0;l<64;l++) for(int j=0;j<4;j++){ int g=l>>4,i=l&15; int exp = mode==0 ? (g*64 + j*16 + i) : (g*128 + j*32 + i); if (o[l*4+j]!=exp) bad++; }
    printf("ds_read_tr16_b64 mode %d hypothesis (lane(i,g) elem j = block[j][i]): %s (bad=%d)\n", mode, bad?"FAIL":"OK", bad);
    if (bad) { for (int l=0;l<64;l++){ printf("  l%2d:",l); for(int j=0;j<4;j++) printf(" %4d",o[l*4+j]); printf("\n"); } }
  }
  // ---- global_load_lds ----
  {
    std::vector<short> g(1024); for (int i=0;i<1024;i++) g[i]=i; short *dg,*dout; CK(hipMalloc(&dg,2048)); CK(hipMalloc(&dout,1024));
    CK(hipMemcpy(dg,g.data(),2048,hipMemcpyHostToDevice));
    k_glds<<<1,64>>>(dg,dout); CK(hipDeviceSynchronize());
    std::vector<short> o(512); CK(hipMemcpy(o.data(),dout,1024,hipMemcpyDeviceToHost));
    int bad=0; for (int l=0;l<64;l++) for(int e=0;e<8;e++) if (o[l*8+e]!=(63-l)*8+e) bad++;
    printf("global_load_lds 16B (dest = base + lane*16, source per-lane): %s (bad=%d)\n", bad?"FAIL":"OK", bad);
    // buffer lds OOB
    k_buflds<<<1,64>>>(dg,dout, 40*16); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(),dout,1024,hipMemcpyDeviceToHost));
    int inb=0, z=0, m1=0, other=0; for (int l=0;l<64;l++) for(int e=0;e<8;e++){ short v=o[l*8+e]; if (l<40){ if (v!=l*8+e) inb++; } else { if (v==0) z++; else if (v==-1) m1++; else other++; } }
    printf("buffer_load_lds OOB (40 of 64 lanes in range): in-range bad=%d; OOB lanes: zeros=%d untouched(-1)=%d other=%d (of %d)\n", inb, z, m1, other, 24*8);
    int* di; CK(hipMalloc(&di,1024)); k_bufreg<<<1,64>>>(dg,di,40*16); CK(hipDeviceSynchronize());
    std::vector<int> oi(256); CK(hipMemcpy(oi.data(),di,1024,hipMemcpyDeviceToHost));
    int zr=0,nz=0; for (int l=40;l<64;l++) for(int e=0;e<4;e++) { if (oi[l*4+e]==0) zr++; else nz++; }
    printf("buffer_load (regs) OOB: zeros=%d nonzero=%d\n", zr, nz);
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device: %s CUs=%d clock=%d kHz memclock=%d kHz L2=%d\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize);
  return 0;
}
