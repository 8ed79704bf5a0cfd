#include "../../cna_amd/csrc/host_eig.c"
#include <stdio.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec*1e3+t.tv_nsec*1e-6;}
int main(int argc,char**argv){
  int n=argc>1?atoi(argv[1]):200,k=argc>2?atoi(argv[2]):16;
  double*G=malloc(8*n*n);
  srand(1);
  /* G = B B^T / n with decaying column scale */
  double*B=malloc(8*n*n);
  for(int i=0;i<n;i++)for(int j=0;j<n;j++)B[i*n+j]=((rand()/(double)RAND_MAX)-0.5)/(1.0+0.15*j);
  for(int i=0;i<n;i++)for(int j=0;j<n;j++){double s=0;for(int c=0;c<n;c++)s+=B[i*n+c]*B[j*n+c];G[i*n+j]=s;}
  size_t nn=(size_t)n*n;
  double*A=malloc(8*(2*nn+10*n+(k+1)*n));double*V=A+nn,*d=V+nn,*e=d+n,*tau=e+n,*work=tau+n,*Y=work+6*n;
  double lam[MAXT],norm1;
  double t[6]={0};int reps=50;
  for(int r=0;r<reps;r++){
    double t0=now();memcpy(A,G,8*nn);memset(tau,0,8*n);e[n-1]=0;
    tridiagonalise(A,n,d,e,V,tau,work);double t1=now();
    largest_eigenvalues(d,e,n,k+1,lam,&norm1);double t2=now();
    tridiagonal_vectors(d,e,n,k,lam,norm1,Y);double t3=now();
    double rr,oo;check_tridiagonal_pairs(d,e,n,k,Y,lam,&rr,&oo);double t5a=now();
    apply_reflectors(V,tau,n,k,Y);double t4=now();
    double t5=t4+(t5a-t3); t4-= (t5a-t3);
    t[0]+=t1-t0;t[1]+=t2-t1;t[2]+=t3-t2;t[3]+=t4-t3;t[4]+=t5-t4;
  }
  printf("n=%d k=%d tridiag %.3f bisect %.3f invit %.3f backtr %.3f check %.3f ms\n",n,k,t[0]/reps,t[1]/reps,t[2]/reps,t[3]/reps,t[4]/reps);
}
