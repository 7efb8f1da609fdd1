#include <cstdio>
#include <cstdint>
#include <chrono>
#include <vector>
#include <string>
#include "fokl_hip.h"
void fokl_set_global_error(const std::string&) {}
int main(){ int D=2000; for (int p : {2,60}) { std::vector<uint32_t> key(624); for(int i=0;i<624;i++) key[i]=i*2654435761u+1; int32_t pos=624, hg=0; double c=0;
 std::vector<double> nm((size_t)D*p+16), r2((size_t)D*(p/2+1)+8), g1(D), g2(D); std::vector<int32_t> lead(D);
 for(int rep=0;rep<2;rep++){ auto t=std::chrono::steady_clock::now(); for(int i=0;i<50;i++) fokl_noise_tape(p,D,5e5,30.0,key.data(),&pos,&hg,&c,nm.data(),r2.data(),lead.data(),g1.data(),g2.data(),nullptr); double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t).count()/50; if(rep) printf("p=%d ns/iter %.1f\n", p, dt/D*1e9); } } }
