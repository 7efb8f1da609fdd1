// Development aid: ns per Gibbs iteration of fokl_noise_tape through the built library (usage: tape_bench <path to libfokl_hip.so>;
// FOKL_SAMPLER_ISA=base forces the portable recorder).  g++ -O2 -std=c++17 tools/tape_bench.cpp -o tape_bench -ldl
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <vector>
#include <dlfcn.h>
typedef int (*tape_fn)(int,int,double,double,uint32_t*,int32_t*,int32_t*,double*,double*,double*,int32_t*,double*,double*,int32_t*);
int main(int argc,char**argv){ void* h=dlopen(argv[1],RTLD_NOW); if(!h){printf("%s\n",dlerror());return 1;} tape_fn f=(tape_fn)dlsym(h,"fokl_noise_tape"); int D=2000;
 for (int p : {2,60,120}) { std::vector<uint32_t> key(624); for(int i=0;i<624;i++) key[i]=i*2654435761u+1; int32_t pos=624, hg=0; double c=0;
 std::vector<double> nm((size_t)D*p + 16), r2((size_t)D*(p/2+1) + 8), g1(D), g2(D); std::vector<int32_t> lead(D);
 for(int rep=0;rep<2;rep++){ auto t=std::chrono::steady_clock::now(); for(int i=0;i<50;i++) f(p,D,5e5,30.0,key.data(),&pos,&hg,&c,nm.data(),r2.data(),lead.data(),g1.data(),g2.data(),nullptr); double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t).count()/50; printf("p=%d ns/iter %.1f\n", p, dt/D*1e9); } } }
