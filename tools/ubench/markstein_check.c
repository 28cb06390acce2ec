// K7a divides by a divisor that is used many times with one true division for y = RN(1 / d) and, per quotient, q = RN(a y); r = fma(-q, d, a);
// RN(fma(r, y, q)) (Markstein).  This compares the sequence with `/` on 208 divisors x 3e6 numerators and, beside it, sqrt(x * x) with |x|
// (Huber without sqrt): `gcc -O2 -mfma -ffp-contract=off markstein_check.c -lm && ./a.out` prints "bad div 0 bad sqrt 0".
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
static uint64_t s=88172645463325252ULL;
static inline uint64_t rnd(){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
static inline double urand(){return (rnd()>>11)*(1.0/9007199254740992.0);}
int main(){
  double gs[]={0.15,0.10,0.05,0.2,0.108,0.0762,1.0/3,0.12345678};
  long bad=0,bads=0;
  for(int gi=0;gi<8+200;gi++){
    double g= gi<8?gs[gi]:0.01+urand();
    volatile double y=1.0/g;
    for(long k=0;k<3000000;k++){
      double a=(urand()-0.5)*pow(2.0,(int)(rnd()%12)-6);
      double q=a*y; double r=fma(-q,g,a); double q1=fma(r,y,q);
      if(q1!=a/g){bad++; if(bad<5)printf("g=%.17g a=%.17g %.17g %.17g\n",g,a,q1,a/g);}
      double x=a*3.7; if(sqrt(x*x)!=fabs(x)) bads++;
    }
  }
  printf("bad div %ld bad sqrt %ld\n",bad,bads);
}
