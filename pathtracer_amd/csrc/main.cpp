// `pathtracer` command line: same options and option handling as the reference
// (src/application/main.cpp:25-42, 126-187). Rendering runs on the GPU; there is
// no window mode, so -f is required.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "renderer.h"

using namespace ptamd;

static void usage(const char *binaryName) {
  printf("Usage: %s [options] <scenefile>\n", binaryName);
  printf("Program Options:\n");
  printf("  -s  <INT>        Number of camera rays per pixel\n");
  printf("  -l  <INT>        Number of samples per area light\n");
  printf("  -t  <INT>        Number of render threads (accepted; the GPU replaces the worker pool)\n");
  printf("  -m  <INT>        Maximum ray depth\n");
  printf("  -o  <INT>        Accumulate Bounces of Light \n");
  printf("  -e  <PATH>       Path to environment map (accepted and ignored: the environment light is a stub)\n");
  printf("  -b  <FLOAT>      The size of the aperture\n");
  printf("  -d  <FLOAT>      The focal distance\n");
  printf("  -f  <FILENAME>   Image (.png) file to save output to in windowless mode\n");
  printf("  -r  <INT> <INT>  Width and height of output image (if windowless)\n");
  printf("  -p  <x> <y> <dx> <dy> Used with the -f flag to render a cell\n");
  printf("  -c  <FILENAME>   Load camera settings file\n");
  printf("  -a  <INT> <FLOAT> Samples per batch and tolerance for adaptive sampling\n");
  printf("  -H               Enable hemisphere sampling for direct lighting\n");
  printf("  -g  <INT>[,<INT>...]  GPU(s) to render on (default 0; several: pixels dealt round-robin, RCCL gather)\n");
  printf("  -S  <INT>        Seed of the per-pixel random-number generator (default 0)\n");
  printf("  -C               Count BVH node visits and primitive tests (slower)\n");
  printf("  -X               Ideal mirror / refraction / glass BSDFs instead of the reference's black stubs\n");
  printf("  -h               Print this help message\n");
  printf("\n");
}

int main(int argc, char **argv) {
  size_t ns_aa = 1, max_ray_depth = 1, ns_area_light = 1, num_threads = 1, samples_per_batch = 32;
  bool accumulate = true, hemisphere = false, write_to_file = false, counters = false, specular = false;
  float max_tolerance = 0.05f;
  double lensRadius = 0.0, focalDistance = 4.7;
  size_t w = 0, h = 0, x = (size_t)-1, y = 0, dx = 0, dy = 0;
  int device = 0;
  std::vector<int> devices;
  unsigned long long seed = 0;
  std::string output_file_name, cam_settings, sceneFilePath;
  int opt;
  while ((opt = getopt(argc, argv, "s:l:t:m:o:e:hH:f:r:c:b:d:a:p:g:S:CX")) != -1) {
    switch (opt) {
      case 'f': write_to_file = true; output_file_name = std::string(optarg); break;
      case 'r':
        if (optind >= argc) { usage(argv[0]); return 1; }
        w = atoi(argv[optind - 1]);
        h = atoi(argv[optind]);
        optind++;
        break;
      case 'p':
        if (optind + 2 >= argc) { usage(argv[0]); return 1; }
        x = atoi(argv[optind - 1]);
        y = atoi(argv[optind - 0]);
        dx = atoi(argv[optind + 1]);
        dy = atoi(argv[optind + 2]);
        optind += 3;
        break;
      case 's': ns_aa = atoi(optarg); break;
      case 'l': ns_area_light = atoi(optarg); break;
      case 't': num_threads = atoi(optarg); break;
      case 'm': max_ray_depth = atoi(optarg); break;
      case 'o': accumulate = atoi(optarg) > 0; break;
      case 'e': std::cout << "[PathTracer] Environment map " << optarg << " ignored" << std::endl; break;
      case 'c': cam_settings = std::string(optarg); break;
      case 'b': lensRadius = atof(optarg); break;
      case 'd': focalDistance = atof(optarg); break;
      case 'a':
        if (optind >= argc) { usage(argv[0]); return 1; }
        samples_per_batch = atoi(argv[optind - 1]);
        max_tolerance = atof(argv[optind]);
        optind++;
        break;
      case 'H':
        // the reference declares H with an argument and steps optind back
        hemisphere = true;
        optind--;
        break;
      case 'g': {  // one GPU, or several: -g 0,1,2,3
        devices.clear();
        for (const char *q = optarg; *q;) {
          devices.push_back(atoi(q));
          while (*q && *q != ',') q++;
          if (*q == ',') q++;
        }
        if (devices.empty()) { usage(argv[0]); return 1; }
        device = devices[0];
        break;
      }
      case 'S': seed = strtoull(optarg, nullptr, 10); break;
      case 'C': counters = true; break;
      case 'X': specular = true; break;
      default: usage(argv[0]); return 1;
    }
  }
  if (optind >= argc) { usage(argv[0]); return 1; }
  sceneFilePath = argv[optind];
  std::cout << "[PathTracer] Input scene file: " << sceneFilePath << std::endl;
  if (!write_to_file) {
    fprintf(stderr, "[PathTracer] no window mode in this build: give -f <out.png>\n");
    return 1;
  }
  if (samples_per_batch == 0) { fprintf(stderr, "[PathTracer] -a: samples per batch must be positive\n"); return 1; }
  std::string sceneFile = sceneFilePath.substr(sceneFilePath.find_last_of('/') + 1);
  sceneFile = sceneFile.substr(0, sceneFile.find(".dae"));

  // Collada::ColladaParser::load + Application::init/load (main.cpp:204-217)
  ptamd_scene *scene = nullptr;
  if (ptamd_scene_load(sceneFilePath.c_str(), &scene)) {
    fprintf(stderr, "[PathTracer] %s\n", ptamd_last_error());
    return 2;
  }
  RaytracedRenderer renderer(ns_aa, max_ray_depth, accumulate, ns_area_light, 1, 1, 1, num_threads, samples_per_batch,
                             max_tolerance, nullptr, hemisphere, sceneFile, lensRadius, focalDistance, device);
  if (devices.size() > 1 && !renderer.set_devices(devices)) return 2;
  renderer.pt->seed = seed;
  renderer.pt->specular = specular;
  renderer.profile_counters = counters;
  // app->resize(w, h) / load_camera (main.cpp:220-224), then set_up_pathtracer
  // (application.cpp:634-640)
  ptamd_camera cam;
  int rc = ptamd_scene_get_camera(scene, (int32_t)w, (int32_t)h, &cam);
  if (!rc && cam_settings != "") {
    rc = ptamd_scene_load_camera_file(scene, cam_settings.c_str());
    if (!rc) rc = ptamd_scene_get_camera(scene, 0, 0, &cam);
  }
  if (rc) {
    fprintf(stderr, "[PathTracer] %s\n", ptamd_last_error());
    return 2;
  }
  if (!w || !h) { w = 800; h = 600; }  // Application::init default (application.cpp:98)
  renderer.set_camera(&cam);
  renderer.set_scene(scene);
  renderer.set_frame_size(w, h);
  renderer.render_to_file(output_file_name, x, y, dx, dy);
  return renderer.is_done() ? 0 : 3;
}
