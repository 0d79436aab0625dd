/* tools/probes/parse_probe.c: per-line cost of the stages of nsnp_mpileup_parse_into on one core of the GPU box host (tools/parse_probe.sh) */
#include "../../nanosnp_amd/csrc/nsnp_textio.c"
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
__attribute__((target("avx2,avx512f,avx512bw"))) int main(){
  FILE*f=fopen("/tmp/nsnp_parse_probe.mpileup","rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);char*t=malloc(n+64);fread(t,1,n,f);
  const char* end=t+n,*limit=t+n;
  rec_list L={malloc(24*(n/8+1024)),n/8+1024,0,0}; memset(L.r,1,24*(n/8+1024));
  int64_t*pos=malloc(n),*off=malloc(n+8);uint8_t*b=malloc(n+64); memset(pos,1,n);memset(off,1,n);memset(b,1,n);
  double best[5]={9,9,9,9,9}, best512=9, bestl2=9, bestl5=9; long lines=0;
  for(int rep=0;rep<7;++rep){
   double t0=now(); lines=0; const char*p=t;
   while(p<end){const char*nl=scan_byte_avx2(p,end,limit,'\n'); p=nl<end?nl+1:end; ++lines;}
   double t1=now(); L.m=0;L.nb=0;
   tokenise_avx2(t,end,limit,t,&L);
   double t2=now(); L.m=0;L.nb=0;
   tokenise_generic(t,end,&L);
   double t3=now(); L.m=0;L.nb=0;
   if (__builtin_cpu_supports("avx512bw")) tokenise_avx512(t,end,limit,t,&L); else tokenise_avx2(t,end,limit,t,&L);
   double t35=now(); if (t35-t3<best512) best512=t35-t3;
   { L.m=0;L.nb=0; double a=now(); tokenise_lines_avx2(t,end,limit,t,&L); double b2=now(); if (b2-a<bestl2) bestl2=b2-a;
     L.m=0;L.nb=0; a=now(); if (__builtin_cpu_supports("avx512bw")) tokenise_lines_avx512(t,end,limit,t,&L); else tokenise_lines_avx2(t,end,limit,t,&L); b2=now(); if (b2-a<bestl5) bestl5=b2-a; t35=now(); }
   place_avx2(L.r,L.m,0,0,L.nb,limit,pos,off,b);
   double t5=now();
   place_generic(L.r,L.m,0,0,pos,off,b);
   double t6=now();
   double v[5]={t1-t0,t2-t1,t3-t2,t5-t35,t6-t5};
   for(int i=0;i<5;++i) if(v[i]<best[i]) best[i]=v[i];
  }
  printf("lines %ld  nlscan %.1f ns/line  tok_avx2 %.1f  tok_generic %.1f  tok_avx512 %.1f  lines_avx2 %.1f  lines_avx512 %.1f  place_avx2 %.1f  place_generic %.1f\n",lines,best[0]/lines*1e9,best[1]/lines*1e9,best[2]/lines*1e9,best512/lines*1e9,bestl2/lines*1e9,bestl5/lines*1e9,best[3]/lines*1e9,best[4]/lines*1e9);
}
