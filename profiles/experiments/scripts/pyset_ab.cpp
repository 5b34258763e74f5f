// Timing of host_logic.h: pyset_order (the replay of CPython's set order) on the first-occurrence list of the bench's image-like
// content (179 k colours).  g++ -O3 -std=c++17 -pthread -I include tools/bench_scripts/pyset_ab.cpp -o /tmp/pyset && /tmp/pyset
// (round 4 used it to A/B a prefetching variant: profiles/experiments/r04_pyset_prefetch_ab.txt)
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "../../dither_pie_amd/csrc/host_logic.h"
using namespace dp;
int main(){
    // first-occurrence list of a smooth+grain image
    std::mt19937 g(3); std::normal_distribution<double> nd(0,3);
    std::vector<uint8_t> seen(1<<24,0); std::vector<uint8_t> rgb;
    for(int y=0;y<540;y++)for(int x=0;x<960;x++){ double v[3]={80+60*sin(x/300.0)+40*(y/540.0),110+50*cos(y/200.0)+20*sin(x/97.0),160+70*(y/540.0)+10*sin((x+y)/50.0)};
        uint32_t c[3]; for(int k=0;k<3;k++){double t=v[k]+nd(g); c[k]=(uint32_t)std::min(255.0,std::max(0.0,t));}
        uint32_t p=c[0]|c[1]<<8|c[2]<<16; if(!seen[p]){seen[p]=1; rgb.push_back(c[0]);rgb.push_back(c[1]);rgb.push_back(c[2]);}}
    size_t n=rgb.size()/3; printf("distinct %zu\n",n);
    for(int rep=0;rep<5;rep++){
        std::vector<uint32_t> order;
        auto t0=std::chrono::steady_clock::now();
        pyset_order(rgb.data(),n,order);
        auto t1=std::chrono::steady_clock::now();
        uint64_t chk=0; for(size_t i=0;i<order.size();i++) chk=chk*1315423911u+order[i];
        printf("pyset %.2f ms (%zu) chk %llx\n",std::chrono::duration<double,std::milli>(t1-t0).count(),order.size(),(unsigned long long)chk);
    }
}
