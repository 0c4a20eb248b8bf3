"""The FP32 ray-triangle and ray-sphere pre-tests of the traversal kernels (pathtracer_amd/csrc/prim32.h)
against the reference's FP64 tests (src/scene/triangle.cpp:25-44, src/scene/sphere.cpp:11-72), on the CPU: whenever the
pre-test claims certainty it must agree with the FP64 test as g++ evaluates it without FMA
contraction (the reference's build), over random and adversarial configurations: rays through
edges and vertices, origins on the triangle's plane, slivers, grazing rays, rays that end at the
triangle. The share of undecided cases is printed (it costs speed, not correctness)."""
import os
import subprocess

from conftest import ROOT

SRC = r"""
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <random>
#include "prim32.h"
using namespace ptdev;
struct V3 { double x,y,z; };
static V3 sub(V3 a,V3 b){return{a.x-b.x,a.y-b.y,a.z-b.z};}
static V3 add(V3 a,V3 b){return{a.x+b.x,a.y+b.y,a.z+b.z};}
static V3 mul(V3 a,double s){return{a.x*s,a.y*s,a.z*s};}
static V3 cross(V3 u,V3 v){return{u.y*v.z-u.z*v.y,u.z*v.x-u.x*v.z,u.x*v.y-u.y*v.x};}
static double dot(V3 u,V3 v){return u.x*v.x+u.y*v.y+u.z*v.z;}
static float f_down(double x){float f=(float)x; return ((double)f>x)?nextafterf(f,-INFINITY):f;}
static float f_up(double x){float f=(float)x; return ((double)f<x)?nextafterf(f,INFINITY):f;}
// reference test (triangle.cpp:25-44), without the max_t clause
struct R64 { bool degenerate, uv_ok, tmin_ok; double t,u,v; };
static R64 ref(V3 o,V3 d,V3 p1,V3 e1,V3 e2,double min_t){
  R64 r{}; V3 n=cross(e1,e2);
  if(dot(n,d)==0){r.degenerate=true;return r;}
  V3 s=sub(o,p1),s1=cross(d,e2),s2=cross(s,e1);
  double rse1=1/(dot(s1,e1));
  r.t=dot(s2,e2)*rse1; r.u=dot(s1,s)*rse1; r.v=dot(s2,d)*rse1;
  r.uv_ok=!(r.u<0||r.v<0||r.u+r.v>1);
  r.tmin_ok=!(r.t<min_t);
  return r;
}
// reference sphere (sphere.cpp:11-72): 0 = no candidate, else t = the first root not below min_t
static bool ref_sphere(V3 o,V3 d,V3 c,double r2,double min_t,double&t){
  V3 om=sub(o,c);
  double a=dot(d,d),b=2*dot(om,d),cc=dot(om,om)-r2;
  double discr=b*b-4*a*cc;
  if(discr<0)return false;
  double r2a=1/(2*a);
  double b2a=-b*r2a,sq=std::sqrt(discr)*r2a;
  double t1=b2a-sq,t2=b2a+sq;
  if(t1>=min_t){t=t1;return true;}
  if(t2>=min_t){t=t2;return true;}
  return false;
}
static double SLO=-3,SHI=1,OLO=-2,OHI=2;
int main(int argc,char**argv){ if(argc>5){SLO=atof(argv[2]);SHI=atof(argv[3]);OLO=atof(argv[4]);OHI=atof(argv[5]);}
  long N=argc>1?atol(argv[1]):20000000;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0,1);
  auto rnd=[&](double a,double b){return a+(b-a)*U(rng);};
  long cnt[12][3]={}; long bad=0;
  for(long it=0;it<N;it++){
    int mode=it%12;
    double scale=std::pow(10.0,rnd(SLO,SHI));
    double off=std::pow(10.0,rnd(OLO,OHI));
    V3 c{rnd(-off,off),rnd(-off,off),rnd(-off,off)};
    V3 dir{rnd(-1,1),rnd(-1,1),rnd(-1,1)};
    double nl=std::sqrt(dot(dir,dir)); if(nl<1e-3)continue; dir=mul(dir,1/nl);
    if(it&64){ float q=(float)dir.x; (void)q; }                       // (directions are FP64 unit vectors)
    if((it&128)&&mode!=6){ int k=(it>>8)%3; double*dd=&dir.x; dd[k]=0; nl=std::sqrt(dot(dir,dir)); if(nl<1e-3)continue; dir=mul(dir,1/nl);} // an axis-parallel component
    double min_t=(it&32)?1e-11:0.1;
    float df[3]={(float)dir.x,(float)dir.y,(float)dir.z};
    if(mode>=8){
      // ---- spheres: rays aimed at / near / past the sphere, origins outside, inside, on it
      double r=scale;
      double r2=r*r;
      double dist=std::pow(10.0,rnd(-3,0.5));
      V3 tgt=add(c,mul(V3{rnd(-1,1),rnd(-1,1),rnd(-1,1)}, r*(mode==9?1.0000001:1.2)));
      if(mode==9){ V3 q=sub(tgt,c); double l=std::sqrt(dot(q,q)); if(l==0)continue; tgt=add(c,mul(q,r*(1+rnd(-1,1)*1e-6)/l)); // grazing: a point on the silhouette
        V3 q2=sub(tgt,c); double k=dot(dir,q2)/dot(q2,q2); dir=sub(dir,mul(q2,k)); nl=std::sqrt(dot(dir,dir)); if(nl<1e-6)continue; dir=mul(dir,1/nl); df[0]=(float)dir.x;df[1]=(float)dir.y;df[2]=(float)dir.z; }
      V3 o=sub(tgt,mul(dir,dist));
      if(mode==10){ V3 q=sub(tgt,c); double l=std::sqrt(dot(q,q)); if(l==0)continue; o=add(c,mul(q,r/l)); }   // origin on the sphere (a bounce / shadow ray leaving it)
      if(mode==11){ o=add(c,mul(V3{rnd(-1,1),rnd(-1,1),rnd(-1,1)},0.5*r)); }                                    // origin inside
      double max_t=INFINITY;
      if(it&256){ double tt; if(ref_sphere(o,dir,c,r2,min_t,tt)) max_t=tt*(1+rnd(-1,1)*std::pow(10.0,rnd(-9,-5))); }
      // the walk moves the origin to where the ray enters the sphere's box: some t0 at or before the sphere
      double tt0; float t0=0;
      if((it&1024)&&ref_sphere(o,dir,c,r2,-1e300,tt0)&&tt0>0) t0=(float)(tt0*rnd(0.5,1.0));
      float big=(float)(1.01*(std::fabs(o.x)+std::fabs(o.y)+std::fabs(o.z)))+std::fabs(t0);
      V3 om=add(o,mul(dir,(double)t0));
      float sv[3]={(float)(om.x-c.x),(float)(om.y-c.y),(float)(om.z-c.z)};
      float lo=0,hi=0; float best=f_up(max_t);
      int cls=classify_sphere32(sv,df,(float)r2,f_down(min_t),f_up(min_t),best,t0,big,lo,hi);
      cnt[mode][cls]++;
      double t; bool ok=ref_sphere(o,dir,c,r2,min_t,t);
      if(cls==PT32_MISS){ if(ok&&!(t>(double)best)){ if(bad++<20) printf("BAD SPHERE MISS mode %d t=%.17g best=%g\n",mode,t,best);} }
      else if(cls==PT32_HIT){ if(!ok||!(t>=(double)lo&&t<=(double)hi)){ if(bad++<20) printf("BAD SPHERE HIT mode %d ok=%d t=%.17g [%g,%g]\n",mode,ok,t,lo,hi);} }
      continue;
    }
    V3 p1=add(c,V3{rnd(-1,1)*scale,rnd(-1,1)*scale,rnd(-1,1)*scale});
    V3 p2=add(c,V3{rnd(-1,1)*scale,rnd(-1,1)*scale,rnd(-1,1)*scale});
    V3 p3=add(c,V3{rnd(-1,1)*scale,rnd(-1,1)*scale,rnd(-1,1)*scale});
    if(mode==5){ // sliver
      p3=add(p1,add(mul(sub(p2,p1),rnd(0,1)),V3{rnd(-1,1)*scale*1e-5,rnd(-1,1)*scale*1e-5,rnd(-1,1)*scale*1e-5}));
    }
    // scene data passes through float in the loader for meshes (not always): mix
    if(it&8){p1={(double)(float)p1.x,(double)(float)p1.y,(double)(float)p1.z};p2={(double)(float)p2.x,(double)(float)p2.y,(double)(float)p2.z};p3={(double)(float)p3.x,(double)(float)p3.y,(double)(float)p3.z};}
    V3 e1=sub(p2,p1),e2=sub(p3,p1);
    // target point on / near the triangle
    double bu=rnd(-0.2,1.2),bv=rnd(-0.2,1.2);
    if(mode==1){ bu=rnd(0,1); bv=(it&16)?0.0:1-bu; }            // exactly on an edge
    if(mode==2){ bu=(it&16)?0:1; bv=0; }                         // a vertex
    if(mode==3){ bu=rnd(0,1)+rnd(-1,1)*1e-7; bv=rnd(-1,1)*1e-7; } // very near an edge
    V3 tgt=add(p1,add(mul(e1,bu),mul(e2,bv)));
    double dist=std::pow(10.0,rnd(-3,0.5));
    if(mode==6){ // grazing
      V3 n=cross(e1,e2); double nn=std::sqrt(dot(n,n)); if(nn>0){n=mul(n,1/nn); double k=dot(dir,n); dir=sub(dir,mul(n,k*(1-rnd(0,1)*1e-4))); nl=std::sqrt(dot(dir,dir)); if(nl<1e-6)continue; dir=mul(dir,1/nl);}
      df[0]=(float)dir.x;df[1]=(float)dir.y;df[2]=(float)dir.z; }
    V3 o=sub(tgt,mul(dir,dist));
    double max_t=INFINITY;
    if(mode==4){ o=tgt; }                        // origin on the triangle plane (self / coplanar neighbour)
    if(mode==7){ max_t=dist*(1+rnd(-1,1)*std::pow(10.0,rnd(-9,-5))); } // ray ends at (about) the triangle: the shadow ray at the light
    // the walk moves the origin to where the ray enters the triangle's box: some t0 at or before the plane
    float t0=0;
    if((it&1024)&&mode!=4) t0=(float)(dist*rnd(0.5,1.0));
    float big=(float)(1.01*(std::fabs(o.x)+std::fabs(o.y)+std::fabs(o.z)))+std::fabs(t0);
    V3 s=sub(V3{std::fma((double)t0,dir.x,o.x),std::fma((double)t0,dir.y,o.y),std::fma((double)t0,dir.z,o.z)},p1);
    float sv[3]={(float)s.x,(float)s.y,(float)s.z};
    s=sub(o,p1);
    float e1f[3]={(float)e1.x,(float)e1.y,(float)e1.z},e2f[3]={(float)e2.x,(float)e2.y,(float)e2.z};
    float lo=0,hi=0; float best=f_up(max_t);
    float dmin=INFINITY;
    if(mode==4){ // the ray's own triangle: the kernel that made the ray passes |T| of the reference's test (ptamd_device.hip, k_shade)
      double Tn=dot(cross(s,e1),e2);
      float tu=fmaxf(f_up(std::fabs(Tn)),7.888609052210118e-31f);
      dmin=(tu*1.002f)/f_down(min_t);
      if(it&512){ // an origin a few ulps off the plane, as o = ro + t rd leaves it
        o=add(o,mul(cross(e1,e2),rnd(-1,1)*1e-15/std::max(std::sqrt(dot(cross(e1,e2),cross(e1,e2))),1e-300)));
        s=sub(o,p1); sv[0]=(float)s.x;sv[1]=(float)s.y;sv[2]=(float)s.z;
        Tn=dot(cross(s,e1),e2); tu=fmaxf(f_up(std::fabs(Tn)),7.888609052210118e-31f); dmin=(tu*1.002f)/f_down(min_t);
      }
    }
    int cls=classify_tri32(sv,df,e1f,e2f,f_down(min_t),f_up(min_t),best,dmin,t0,big,lo,hi);
    cnt[mode][cls]++;
    R64 r=ref(o,dir,p1,e1,e2,min_t);
    bool ok64=!r.degenerate&&r.uv_ok&&r.tmin_ok;   // max_t aside
    if(cls==PT32_MISS){
      if(ok64 && !(r.t>(double)best)){ if(bad++<20) printf("BAD MISS mode %d t=%.17g u=%.17g v=%.17g best=%g\n",mode,r.t,r.u,r.v,best);}
    } else if(cls==PT32_HIT){
      if(!ok64 || !(r.t>=(double)lo && r.t<=(double)hi) || !(r.t>min_t)){ if(bad++<20) printf("BAD HIT mode %d t=%.17g [%g,%g] u=%.17g v=%.17g deg=%d\n",mode,r.t,lo,hi,r.u,r.v,r.degenerate);}
    }
  }
  const char*names[12]={"random","on edge","vertex","near edge","origin on plane","sliver","grazing","ends at triangle",
                        "sphere","sphere grazing","sphere from surface","sphere from inside"};
  for(int m=0;m<12;m++){ long t=cnt[m][0]+cnt[m][1]+cnt[m][2]; if(!t)continue; printf("%-20s miss %6.3f%% hit %6.3f%% unsure %6.3f%%\n",names[m],100.0*cnt[m][0]/t,100.0*cnt[m][1]/t,100.0*cnt[m][2]/t);}
  printf("bad %ld\n",bad);
  return bad?1:0;
}

"""


def test_fp32_pretest_never_contradicts_the_fp64_test(tmp_path):
    src = tmp_path / "prim32_fuzz.cpp"
    src.write_text(SRC)
    exe = tmp_path / "prim32_fuzz"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "pathtracer_amd", "csrc"), str(src), "-o", str(exe)])
    for args in (["3000000"], ["2000000", "-2.5", "-1.5", "-0.5", "0.3"], ["2000000", "-1", "0.3", "-0.5", "0.3"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True)
        print(out.stdout)
        assert out.returncode == 0, out.stdout
        assert "bad 0" in out.stdout
