/* Test infrastructure: checks the binary64 sin/cos sequence used by the HIP kernel
 * (orbslamm_amd/csrc/orbx_kernels.hip: sincos_0_2pi -- Cody-Waite reduction by pi/2 + fdlibm kernel polynomials,
 * every operation separately rounded) against the host libm, the way the reference uses it
 * (ORBextractor.cc:112-113: (float)cos((double)angle), (float)sin((double)angle)).
 * usage: sincos_check [step]   step 1 = every binary32 argument in [0, 2pi] (1.09e9 values, ~15 s): bad=0.
 * build: gcc -O2 -ffp-contract=off -o sincos_check sincos_check.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline double ksin(double x){ const double S1=-1.66666666666666324348e-01,S2=8.33333333332248946124e-03,S3=-1.98412698298579493134e-04,S4=2.75573137070700676789e-06,S5=-2.50507602534068634195e-08,S6=1.58969099521155010221e-10;
 double z=x*x; double r=S2+z*(S3+z*(S4+z*(S5+z*S6))); return x+x*z*(S1+z*r);} 
static inline double kcos(double x){ const double C1=4.16666666666666019037e-02,C2=-1.38888888888741095749e-03,C3=2.48015872894767294178e-05,C4=-2.75573143513906633035e-07,C5=2.08757232129817482790e-09,C6=-1.13596475577881948265e-11;
 double z=x*x; double r=z*(C1+z*(C2+z*(C3+z*(C4+z*(C5+z*C6))))); return 1.0-(0.5*z-z*r);} 
static inline void mysincos(double x,double*s,double*c){ const double invpio2=6.36619772367581382433e-01,p1=1.57079632673412561417e+00,p1t=6.07710050650619224932e-11;
 int k=(int)(x*invpio2+0.5); double kd=(double)k; double r=(x-kd*p1)-kd*p1t; double sr=ksin(r),cr=kcos(r);
 switch(k&3){case 0:*s=sr;*c=cr;break;case 1:*s=cr;*c=-sr;break;case 2:*s=-sr;*c=-cr;break;default:*s=-cr;*c=sr;} }
int main(int argc,char**argv){ float hi=6.2831860f; uint32_t hb; memcpy(&hb,&hi,4); uint32_t lo=0, step=1; if(argc>1) step=atoi(argv[1]);
 long bad=0,n=0; for(uint32_t b=lo;b<=hb;b+=step){ float f; memcpy(&f,&b,4); double x=f; double s,c; mysincos(x,&s,&c);
  float fs=(float)s, fc=(float)c, gs=(float)sin(x), gc=(float)cos(x); n++; if(memcmp(&fs,&gs,4)||memcmp(&fc,&gc,4)){ if(bad<10) printf("x=%.9g mine %.9g %.9g glibc %.9g %.9g\n",f,fs,fc,gs,gc); bad++; } }
 printf("n=%ld bad=%ld\n",n,bad); return 0; }
