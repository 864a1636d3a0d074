// exhaustive check: Markstein division q1 = fma(r, y, q0), y = RN(1/c), q0 = RN(x*y), r = fma(-c, q0, x)  ==  x / c (IEEE RN)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
int main(int argc,char**argv){
  int c0=atoi(argv[1]), c1=atoi(argv[2]);
  long bad_total=0;
  #pragma omp parallel for schedule(dynamic,8) reduction(+:bad_total)
  for(int c=c0;c<=c1;c++){
    const float cf=(float)c; const float y=1.0f/cf; // RN(1/c): IEEE division is correctly rounded
    long bad=0;
    for(uint32_t m=0;m<(1u<<23);m++){
      uint32_t bits=0x3f800000u|m; float x; memcpy(&x,&bits,4);
      const float q0=x*y; const float r=fmaf(-cf,q0,x); const float q1=fmaf(r,y,q0);
      const float want=x/cf;
      if(q1!=want){ if(bad<3) printf("c=%d x=%a got %a want %a\n",c,x,q1,want); bad++; }
    }
    bad_total+=bad;
  }
  printf("c in [%d,%d]: mismatches %ld\n",c0,c1,bad_total);
  return 0;
}
