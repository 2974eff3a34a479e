// simulation: relocate "quiet rows" (don't-look bits with exact margin) statistics on the reference's GLS trajectory
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define DX(D,n,a,b) ((D)[(size_t)(a)*(n)+(b)])
static int isclose0(double d){double ad=fabs(d);double r=1e-8+1e-5*ad;return ad<=r;}
static double two_opt_cost(const int*t,const double*D,int n,int i,int j){if(i==j)return 0;if(j<i){int x=i;i=j;j=x;}int a=t[i],b=t[i-1],c=t[j],d=t[j-1];double v=DX(D,n,a,c)+DX(D,n,b,d);v=v-DX(D,n,a,b);v=v-DX(D,n,c,d);return v;}
static void two_opt(int*t,int i,int j){if(j<i){int x=i;i=j;j=x;}int lo=i,hi=j-1;while(lo<hi){int x=t[lo];t[lo]=t[hi];t[hi]=x;lo++;hi--;}}
static double reloc_cost(const int*t,const double*D,int n,int i,int j){if(i==j)return 0;int a=t[i-1],b=t[i],c=t[i+1];int d,e;if(i<j){d=t[j];e=t[j+1];}else{d=t[j-1];e=t[j];}double v=-DX(D,n,a,b);v=v-DX(D,n,b,c);v=v+DX(D,n,a,c);v=v-DX(D,n,d,e);v=v+DX(D,n,d,b);v=v+DX(D,n,b,e);return v;}
static void relocate(int*t,int i,int j){int node=t[i];if(i<j){for(int p=i;p<j;++p)t[p]=t[p+1];}else{for(int p=i;p>j;--p)t[p]=t[p-1];}t[j]=node;}
static int two_opt_a2a(const int*t,const double*D,int n,double*bd,int*bi,int*bj){double best=0;int f=0;for(int i=1;i<=n-1;++i)for(int j=i+1;j<=n-1;++j){if(abs(i-j)<2)continue;double d=two_opt_cost(t,D,n,i,j);if(d<best&&!isclose0(d)){best=d;*bi=i;*bj=j;f=1;}}*bd=best;return f;}
// quiet-row state
static unsigned char active[512];
static long hist_active[600]; static long n_scans_by_idx[64], sum_active_by_idx[64], gt64_by_idx[64];
static long mism=0;
#define THR (-0.5e-8)
// full relocate a2a restricted to rows in `act` (all rows if act==NULL); also records per-row quietness into newq if given
static int reloc_a2a(const int*t,const double*D,int n,const unsigned char*act,unsigned char*newact,double*bd,int*bi,int*bj){double best=0;int f=0;for(int i=1;i<=n-1;++i){int b=t[i];if(act&&!act[b])continue;int rowact=0;for(int j=1;j<=n-1;++j){if(i==j)continue;double d=reloc_cost(t,D,n,i,j);
 if(i-j==1){ if(d<THR)rowact=1; continue;} // orientation-excluded candidate counts for the quiet test only
 if(d<THR)rowact=1; if(d<best&&!isclose0(d)){best=d;*bi=i;*bj=j;f=1;}}
 if(newact)newact[b]=rowact;}
 *bd=best;return f;}
static void touch(int b){active[b]=1;}
// after a move: edges added (x,y) list; nodes touched
static void check_new_edge(const int*t,const double*D,int n,int x,int y){ // for every row b not adjacent: delta of inserting b between x,y (both orders)
 for(int i=1;i<=n-1;++i){int b=t[i];if(active[b])continue;if(b==x||b==y)continue;int a=t[i-1],c=t[i+1];double base=-DX(D,n,a,b);base=base-DX(D,n,b,c);base=base+DX(D,n,a,c);double d1=((base-DX(D,n,x,y))+DX(D,n,x,b))+DX(D,n,b,y);double d2=((base-DX(D,n,y,x))+DX(D,n,y,b))+DX(D,n,b,x);if(d1<THR||d2<THR)active[b]=1;}}
static int ls(int*t,double*cost,const double*D,int n,int use_bits_from_start){
 int moves=0,improved=1,ridx=0;int have_bits=use_bits_from_start;
 while(improved){improved=0;for(int op=0;op<2;++op){double delta;int bi=0,bj=0,f;
  if(op==0){f=two_opt_a2a(t,D,n,&delta,&bi,&bj);}
  else{
   double fd;int fi=0,fj=0;int ff=reloc_a2a(t,D,n,NULL,NULL,&fd,&fi,&fj); // reference result
   if(have_bits){int A=0;for(int b=1;b<n;++b)A+=active[b];hist_active[A]++;if(ridx<64){n_scans_by_idx[ridx]++;sum_active_by_idx[ridx]+=A;if(A>64)gt64_by_idx[ridx]++;}
     unsigned char na[512];memcpy(na,active,sizeof na);double qd;int qi=0,qj=0;int qf=reloc_a2a(t,D,n,active,na,&qd,&qi,&qj);
     if(qf!=ff||(ff&&(qd!=fd||qi!=fi||qj!=fj)))mism++;
     memcpy(active,na,sizeof na);
   } else { unsigned char na[512];memset(na,0,sizeof na);double qd;int qi,qj;reloc_a2a(t,D,n,NULL,na,&qd,&qi,&qj);memcpy(active,na,sizeof na);have_bits=1; if(ridx<64){n_scans_by_idx[ridx]++;sum_active_by_idx[ridx]+=n-1;gt64_by_idx[ridx]++;}}
   ridx++; f=ff;delta=fd;bi=fi;bj=fj;}
  if(f&&delta<0){improved=1;*cost+=delta;
   if(op==0){int i=bi<bj?bi:bj,j=bi<bj?bj:bi;int a=t[i],b=t[i-1],c=t[j],d=t[j-1];two_opt(t,i,j);touch(a);touch(b);touch(c);touch(d);
     // reversal flips pred/succ of interior nodes: handled by counting the orientation-excluded candidate in the quiet test
     if(have_bits){check_new_edge(t,D,n,b,d);check_new_edge(t,D,n,a,c);}}
   else{int i=bi,j=bj;int a=t[i-1],b=t[i],c=t[i+1];int d,e;if(i<j){d=t[j];e=t[j+1];}else{d=t[j-1];e=t[j];}relocate(t,i,j);touch(a);touch(b);touch(c);touch(d);touch(e);
     if(have_bits){check_new_edge(t,D,n,a,c);check_new_edge(t,D,n,d,b);check_new_edge(t,D,n,b,e);}}
   moves++;}}}
 return moves;}
int main(int argc,char**argv){int n=argc>1?atoi(argv[1]):100;int iters=argc>2?atoi(argv[2]):300;int gmode=argc>3?atoi(argv[3]):0;int carry=argc>4?atoi(argv[4]):0;unsigned seed=argc>5?atoi(argv[5]):1;srand(seed);
 double*px=malloc(n*sizeof(double)),*py=malloc(n*sizeof(double));for(int i=0;i<n;++i){px[i]=rand()/(double)RAND_MAX;py[i]=rand()/(double)RAND_MAX;}
 double*D=malloc(sizeof(double)*n*n),*G=malloc(sizeof(double)*n*n),*pen=calloc(n*n,sizeof(double)),*Dg=malloc(sizeof(double)*n*n);
 for(int i=0;i<n;++i)for(int j=0;j<n;++j){DX(D,n,i,j)=hypot(px[i]-px[j],py[i]-py[j]);}
 for(int i=0;i<n;++i)for(int j=i;j<n;++j){double g=gmode==0?DX(D,n,i,j):(gmode==1?(float)(rand()/(double)RAND_MAX>0.5?rand()/(double)RAND_MAX:0.0):(float)(DX(D,n,i,j)*0.5+0.3*rand()/(double)RAND_MAX));DX(G,n,i,j)=g;DX(G,n,j,i)=g;}
 int*t=malloc(sizeof(int)*(n+1));// nearest neighbour tour on G
 {char*vis=calloc(n,1);t[0]=0;vis[0]=1;for(int l=1;l<n;++l){int i=t[l-1],bj=-1;double bw=0;for(int j=0;j<n;++j){if(j==i||vis[j])continue;double w=DX(G,n,i,j);if(bj<0||w<bw){bj=j;bw=w;}}t[l]=bj;vis[bj]=1;}t[n]=0;}
 double cost=0;for(int p=0;p<n;++p)cost+=DX(D,n,t[p],t[p+1]);double k=0.1*cost/n;memcpy(Dg,D,sizeof(double)*n*n);
 memset(active,1,sizeof active);
 ls(t,&cost,D,n,0);
 memset(n_scans_by_idx,0,sizeof n_scans_by_idx);memset(sum_active_by_idx,0,sizeof sum_active_by_idx);memset(gt64_by_idx,0,sizeof gt64_by_idx);memset(hist_active,0,sizeof hist_active);
 long total_scans=0;long pert_touched_sum=0;
 for(int it=0;it<iters;++it){int moves=0;
  while(moves<20){double mu=0;int me=-1;for(int p=0;p<n;++p){int u=t[p],v=t[p+1];double ut=DX(G,n,u,v)/(1.0+DX(pen,n,u,v));if(ut>mu||me<0){mu=ut;me=p;}}
   int eu=t[me],ev=t[me+1];DX(pen,n,eu,ev)+=1;DX(pen,n,ev,eu)=DX(pen,n,eu,ev);double kp=k*DX(pen,n,eu,ev);DX(Dg,n,eu,ev)=DX(D,n,eu,ev)+kp;DX(Dg,n,ev,eu)=DX(D,n,ev,eu)+kp;
   int ends[2]={eu,ev};for(int s=0;s<2;++s){int node=ends[s];if(node==0)continue;int i=0;while(t[i]!=node)++i;for(int op=0;op<2;++op){double best=0;int bj=0,f=0;for(int j=1;j<=n-1;++j){if(op==0){if(abs(i-j)<2)continue;}else{if(i==j)continue;}double d=op==0?two_opt_cost(t,Dg,n,i,j):reloc_cost(t,Dg,n,i,j);if(d<best&&!isclose0(d)){best=d;bj=j;f=1;}}
     if(f&&best<0){if(op==0){int ii=i<bj?i:bj,jj=i<bj?bj:i;int a=t[ii],b=t[ii-1],c=t[jj],d=t[jj-1];two_opt(t,ii,jj);touch(a);touch(b);touch(c);touch(d);if(carry){check_new_edge(t,D,n,b,d);check_new_edge(t,D,n,a,c);}}
       else{int a=t[i-1],b=t[i],c=t[i+1];int d,e;if(i<bj){d=t[bj];e=t[bj+1];}else{d=t[bj-1];e=t[bj];}relocate(t,i,bj);touch(a);touch(b);touch(c);touch(d);touch(e);if(carry){check_new_edge(t,D,n,a,c);check_new_edge(t,D,n,d,b);check_new_edge(t,D,n,b,e);}}
      moves++;}}}}
  cost=0;for(int p=0;p<n;++p)cost+=DX(D,n,t[p],t[p+1]);
  if(carry){int A=0;for(int b=1;b<n;++b)A+=active[b];pert_touched_sum+=A;}
  ls(t,&cost,D,n,carry);}
 printf("n=%d iters=%d guide=%d carry=%d mismatches=%ld\n",n,iters,gmode,carry,mism);
 for(int r=0;r<12;++r)if(n_scans_by_idx[r])printf("relocate scan #%d of a descent: count %ld, mean active rows %.1f, >64 active: %.2f\n",r,n_scans_by_idx[r],sum_active_by_idx[r]/(double)n_scans_by_idx[r],gt64_by_idx[r]/(double)n_scans_by_idx[r]);
 if(carry)printf("mean active after perturbation: %.1f\n",pert_touched_sum/(double)iters);
 return 0;}
