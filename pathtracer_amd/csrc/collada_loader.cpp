#include "collada_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>

#include "xml_lite.h"

namespace ptamd {
namespace {

struct LoadError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Whitespace-separated number reader with the behaviour of `stringstream >> x`
// as the reference uses it: once a read fails every later read yields 0.
struct NumStream {
  const char *p;
  bool failed = false;
  explicit NumStream(const char *s) : p(s ? s : "") {}
  void skip() {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') p++;
  }
  double next_double() {
    if (failed) return 0;
    skip();
    char *e;
    double v = strtod(p, &e);
    if (e == p) { failed = true; return 0; }
    p = e;
    return v;
  }
  float next_float() {
    if (failed) return 0;
    skip();
    char *e;
    float v = strtof(p, &e);
    if (e == p) { failed = true; return 0; }
    p = e;
    return v;
  }
  size_t next_size() {
    if (failed) return 0;
    skip();
    char *e;
    unsigned long long v = strtoull(p, &e, 10);
    if (e == p) { failed = true; return 0; }
    p = e;
    return (size_t)v;
  }
};

V3 spectrum_from_string(const char *s) {  // collada.cpp:30-41
  NumStream ns(s);
  V3 v;
  v.x = ns.next_double();
  v.y = ns.next_double();
  v.z = ns.next_double();
  return v;
}

struct Loader {
  SceneInfo *scene = nullptr;
  V3 up;
  M4 transform = M4::identity();
  std::map<std::string, XmlElement *> sources;

  void uri_load(XmlElement *xml) {  // collada.cpp:57-70
    if (const char *id = xml->attribute("id")) sources[id] = xml;
    for (auto &c : xml->children) uri_load(c.get());
  }
  XmlElement *uri_find(const std::string &id) {  // collada.cpp:72-76
    auto it = sources.find(id);
    return it == sources.end() ? nullptr : it->second;
  }
  // '/'-separated descent by first matching child; a trailing element that
  // carries a url attribute is replaced by its target (collada.cpp:79-99)
  XmlElement *get_element(XmlElement *xml, const std::string &query) {
    XmlElement *e = xml;
    size_t b = 0;
    while (e && b <= query.size()) {
      size_t s = query.find('/', b);
      std::string tok = query.substr(b, s == std::string::npos ? std::string::npos : s - b);
      e = e->first_child(tok.c_str());
      if (s == std::string::npos) break;
      b = s + 1;
    }
    if (e) {
      if (const char *url = e->attribute("url")) e = uri_find(url + 1);
    }
    return e;
  }
  XmlElement *get_technique_common(XmlElement *xml) {  // collada.cpp:102-116
    if (XmlElement *cp = xml->first_child("profile_COMMON")) {
      for (XmlElement *t = cp->first_child("technique"); t; t = t->next_sibling("technique")) {
        const char *sid = t->attribute("sid");
        if (sid && !strcmp(sid, "common")) return t;
      }
    }
    return xml->first_child("technique_common");
  }
  XmlElement *get_technique_CGL(XmlElement *xml) {  // collada.cpp:119-130
    for (XmlElement *t = get_element(xml, "extra/technique"); t; t = t->next_sibling("technique")) {
      const char *profile = t->attribute("profile");
      if (profile && !strcmp(profile, "CGL")) return t;
    }
    return nullptr;
  }

  static const char *text_or_throw(XmlElement *e, const char *what) {
    const char *t = e ? e->get_text() : nullptr;
    if (!t) throw LoadError(std::string("missing text in ") + what);
    return t;
  }

  void parse_camera(XmlElement *xml, CameraInfo &camera) {  // collada.cpp:434-475
    camera.up_dir = up;
    camera.view_dir = V3(0, 0, -1);
    XmlElement *persp = get_element(xml, "optics/technique_common/perspective");
    if (!persp) throw LoadError("no perspective defined in camera");
    XmlElement *e_xfov = persp->first_child("xfov");
    XmlElement *e_yfov = persp->first_child("yfov");
    XmlElement *e_znear = persp->first_child("znear");
    XmlElement *e_zfar = persp->first_child("zfar");
    camera.hFov = e_xfov ? (float)atof(text_or_throw(e_xfov, "xfov")) : 50.0f;
    camera.vFov = e_yfov ? (float)atof(text_or_throw(e_yfov, "yfov")) : 35.0f;
    camera.nClip = e_znear ? (float)atof(text_or_throw(e_znear, "znear")) : 0.001f;
    camera.fClip = e_zfar ? (float)atof(text_or_throw(e_zfar, "zfar")) : 1000.0f;
    if (!e_yfov) {
      XmlElement *e_ar = get_element(persp, "aspect_ratio");
      if (!e_ar) throw LoadError("incomplete perspective definition in camera");
      float aspect_ratio = (float)atof(text_or_throw(e_ar, "aspect_ratio"));
      camera.vFov = (float)(2 * degrees(atan(tan(radians(0.5 * camera.hFov)) / aspect_ratio)));
    }
  }

  void parse_light(XmlElement *xml, LightInfo &light) {  // collada.cpp:477-578
    XmlElement *tc = get_technique_common(xml);
    XmlElement *tg = get_technique_CGL(xml);
    XmlElement *technique = tg ? tg : tc;
    if (!technique) throw LoadError("no supported profile defined in light");
    XmlElement *e_light = technique->first_child();
    if (!e_light) return;
    const std::string type = e_light->name;
    XmlElement *e_color = get_element(e_light, "color");
    if (type == "ambient") light.light_type = LIGHT_AMBIENT;
    else if (type == "directional") light.light_type = LIGHT_DIRECTIONAL;
    else if (type == "area") light.light_type = LIGHT_AREA;
    else if (type == "point") {
      light.light_type = LIGHT_POINT;
      if (!get_element(e_light, "constant_attenuation") || !get_element(e_light, "linear_attenuation") ||
          !get_element(e_light, "quadratic_attenuation"))
        throw LoadError("incomplete definition of point light");
    } else if (type == "spot") {
      light.light_type = LIGHT_SPOT;
      if (!e_light->first_child("falloff_angle") || !e_light->first_child("falloff_exponent") ||
          !get_element(e_light, "constant_attenuation") || !get_element(e_light, "linear_attenuation") ||
          !get_element(e_light, "quadratic_attenuation"))
        throw LoadError("incomplete definition of spot light");
    } else {
      throw LoadError("light type " + type + " is not supported");
    }
    if (!e_color) throw LoadError("no color definition in light");
    light.spectrum = spectrum_from_string(text_or_throw(e_color, "light color"));
  }

  void parse_sphere(XmlElement *xml, SphereInfo &sphere) {  // collada.cpp:580-603
    XmlElement *tech = get_technique_CGL(xml);
    if (!tech) throw LoadError("no CGL profile technique in sphere geometry");
    XmlElement *e_radius = get_element(tech, "sphere/radius");
    if (!e_radius) throw LoadError("invalid sphere definition in geometry");
    sphere.radius = (float)atof(text_or_throw(e_radius, "sphere radius"));
  }

  void parse_polymesh(XmlElement *xml, PolymeshInfo &pm) {  // collada.cpp:606-852
    XmlElement *e_mesh = xml->first_child("mesh");
    if (!e_mesh) throw LoadError("no mesh data defined in geometry");
    // float arrays are parsed as float and widened to double later (:631-641)
    std::map<std::string, std::vector<float>> arr;
    for (XmlElement *src = e_mesh->first_child("source"); src; src = src->next_sibling("source")) {
      const char *sid = src->attribute("id");
      XmlElement *fa = src->first_child("float_array");
      if (!sid || !fa) continue;
      const char *cnt = fa->attribute("count");
      size_t n = cnt ? (size_t)atoi(cnt) : 0;
      std::vector<float> floats;
      floats.reserve(n);
      NumStream ns(fa->get_text());
      float f = 0;
      for (size_t i = 0; i < n; i++) {
        // a failed stream read leaves the previous value in place (:639-641)
        bool was_failed = ns.failed;
        float v = ns.next_float();
        if (!was_failed && !ns.failed) f = v;
        else if (!was_failed && ns.failed) f = 0;
        floats.push_back(f);
      }
      arr[sid] = std::move(floats);
    }
    XmlElement *e_vertices = e_mesh->first_child("vertices");
    if (!e_vertices) throw LoadError("no vertices defined in geometry");
    const char *vid = e_vertices->attribute("id");
    std::string vertices_id = vid ? vid : "";
    std::vector<V3> vertices;
    for (XmlElement *in = e_vertices->first_child("input"); in; in = in->next_sibling("input")) {
      const char *sem = in->attribute("semantic");
      if (sem && !strcmp(sem, "POSITION")) {
        const char *src = in->attribute("source");
        std::string source = src ? src + 1 : "";
        auto it = arr.find(source);
        if (it == arr.end()) throw LoadError("undefined input source: " + source);
        const std::vector<float> &fl = it->second;
        for (size_t i = 0; i + 2 < fl.size(); i += 3)
          vertices.push_back(V3(fl[i], fl[i + 1], fl[i + 2]));
      }
    }
    XmlElement *pl = e_mesh->first_child("polylist");
    if (!pl) return;  // reference: no polylist -> no polygons (:689-690)
    bool has_v = false, has_n = false, has_t = false;
    size_t v_off = 0;
    for (XmlElement *in = pl->first_child("input"); in; in = in->next_sibling("input")) {
      const char *sem = in->attribute("semantic");
      const char *src = in->attribute("source");
      const char *off = in->attribute("offset");
      std::string semantic = sem ? sem : "";
      std::string source = src ? src + 1 : "";
      size_t offset = off ? (size_t)atoi(off) : 0;
      if (semantic == "VERTEX") {
        has_v = true;
        v_off = offset;
        if (source != vertices_id) throw LoadError("undefined source for VERTEX semantic: " + source);
        pm.vertices = vertices;
      }
      if (semantic == "NORMAL") {
        has_n = true;
        if (arr.find(source) == arr.end()) throw LoadError("undefined source for NORMAL semantic: " + source);
      }
      if (semantic == "TEXCOORD") {
        has_t = true;
        if (arr.find(source) == arr.end()) throw LoadError("undefined source for TEXCOORD semantic: " + source);
      }
    }
    const char *cnt = pl->attribute("count");
    size_t num_polygons = cnt ? (size_t)atoi(cnt) : 0;
    size_t stride = (has_v ? 1 : 0) + (has_n ? 1 : 0) + (has_t ? 1 : 0);
    XmlElement *e_vcount = pl->first_child("vcount");
    if (!e_vcount) throw LoadError("polygon sizes undefined in geometry");
    std::vector<size_t> sizes;
    size_t num_indices = 0;
    {
      NumStream ns(e_vcount->get_text());
      size_t size = 0;
      for (size_t i = 0; i < num_polygons; i++) {
        size_t v = ns.next_size();
        if (!ns.failed) size = v;
        sizes.push_back(size);
        num_indices += size * stride;
      }
    }
    XmlElement *e_p = pl->first_child("p");
    if (!e_p) throw LoadError("no index array defined in geometry");
    std::vector<size_t> indices;
    indices.reserve(num_indices);
    {
      NumStream ns(e_p->get_text());
      size_t index = 0;
      for (size_t i = 0; i < num_indices; i++) {
        size_t v = ns.next_size();
        if (!ns.failed) index = v;
        indices.push_back(index);
      }
    }
    pm.polygons.resize(num_polygons);
    if (has_v) {
      size_t k = 0;
      for (size_t i = 0; i < num_polygons; i++)
        for (size_t j = 0; j < sizes[i]; j++) {
          size_t at = k * stride + v_off;
          if (at >= indices.size()) throw LoadError("polylist index array too short");
          pm.polygons[i].push_back(indices[at]);
          k++;
        }
    }
  }

  void parse_material(XmlElement *xml, Material &m) {  // collada.cpp:854-938
    XmlElement *e_effect = get_element(xml, "instance_effect");
    if (!e_effect) throw LoadError("no target effects found for material");
    XmlElement *tc = get_technique_common(e_effect);
    XmlElement *tg = get_technique_CGL(e_effect);
    if (tg) {
      bool set = false;
      for (XmlElement *b = tg->first_child(); b; b = b->next_sibling()) {
        const std::string type = b->name;
        if (type == "emission") {
          m = Material();
          m.kind = MAT_EMISSION;
          m.a = spectrum_from_string(text_or_throw(get_element(b, "radiance"), "emission radiance"));
          set = true;
        } else if (type == "mirror") {
          m = Material();
          m.kind = MAT_MIRROR;
          m.a = spectrum_from_string(text_or_throw(get_element(b, "reflectance"), "mirror reflectance"));
          set = true;
        } else if (type == "microfacet") {
          m = Material();
          m.kind = MAT_MICROFACET;
          m.alpha = (float)atof(text_or_throw(get_element(b, "alpha"), "microfacet alpha"));
          m.a = spectrum_from_string(text_or_throw(get_element(b, "eta"), "microfacet eta"));
          m.b = spectrum_from_string(text_or_throw(get_element(b, "k"), "microfacet k"));
          set = true;
        } else if (type == "refraction") {
          m = Material();
          m.kind = MAT_REFRACTION;
          m.a = spectrum_from_string(text_or_throw(get_element(b, "transmittance"), "refraction transmittance"));
          m.alpha = (float)atof(text_or_throw(get_element(b, "roughness"), "refraction roughness"));
          m.ior = (float)atof(text_or_throw(get_element(b, "ior"), "refraction ior"));
          set = true;
        } else if (type == "glass") {
          m = Material();
          m.kind = MAT_GLASS;
          m.a = spectrum_from_string(text_or_throw(get_element(b, "transmittance"), "glass transmittance"));
          m.b = spectrum_from_string(text_or_throw(get_element(b, "reflectance"), "glass reflectance"));
          m.alpha = (float)atof(text_or_throw(get_element(b, "roughness"), "glass roughness"));
          m.ior = (float)atof(text_or_throw(get_element(b, "ior"), "glass ior"));
          set = true;
        }
      }
      // the reference leaves a null BSDF here and crashes when shading
      if (!set) throw LoadError("CGL material technique without a known BSDF");
    } else if (tc) {
      XmlElement *e_diffuse = get_element(tc, "phong/diffuse/color");
      m = Material();
      m.kind = MAT_DIFFUSE;
      if (e_diffuse) m.a = spectrum_from_string(text_or_throw(e_diffuse, "diffuse color"));
      else m.a = V3(.5f, .5f, .5f);
    } else {
      m = Material();
    }
  }

  bool bind_material(XmlElement *node_xml, Material &m) {  // collada.cpp:373-393
    XmlElement *im = get_element(node_xml, "instance_geometry/bind_material/technique_common/instance_material");
    if (!im) return false;
    const char *target = im->attribute("target");
    if (!target) throw LoadError("no target material in instance_material");
    XmlElement *e_material = uri_find(target + 1);
    if (!e_material) throw LoadError(std::string("invalid target material id: ") + (target + 1));
    parse_material(e_material, m);
    return true;
  }

  void parse_node(XmlElement *xml) {  // collada.cpp:236-432
    Node node;
    if (const char *a = xml->attribute("id")) node.id = a;
    if (const char *a = xml->attribute("name")) node.name = a;
    for (XmlElement *e = xml->first_child(); e; e = e->next_sibling()) {
      const std::string name = e->name;
      if (name == "matrix") {
        NumStream ns(e->get_text());
        M4 mat;
        for (int i = 0; i < 4; i++)
          for (int j = 0; j < 4; j++) mat(i, j) = ns.next_double();
        node.transform = mat;
        break;  // a <matrix> ends the transform list (:266)
      }
      if (name == "rotate") {
        // a zero matrix with four entries read from the text (:271-298)
        M4 m;
        NumStream ns(e->get_text());
        const char *sid = e->attribute("sid");
        char axis = (sid && *sid) ? sid[strlen(sid) - 1] : 0;
        switch (axis) {
          case 'X': m(1, 1) = ns.next_double(); m(1, 2) = ns.next_double(); m(2, 1) = ns.next_double(); m(2, 2) = ns.next_double(); break;
          case 'Y': m(0, 0) = ns.next_double(); m(2, 0) = ns.next_double(); m(0, 2) = ns.next_double(); m(2, 2) = ns.next_double(); break;
          case 'Z': m(0, 0) = ns.next_double(); m(0, 1) = ns.next_double(); m(1, 0) = ns.next_double(); m(1, 1) = ns.next_double(); break;
          default: break;
        }
        node.transform = mul_ref(m, node.transform);
      }
      if (name == "translate") {  // (:301-311)
        M4 m;
        NumStream ns(e->get_text());
        m(0, 3) = ns.next_double();
        m(1, 3) = ns.next_double();
        m(2, 3) = ns.next_double();
        node.transform = mul_ref(m, node.transform);
      }
      if (name == "scale") {  // (:314-324) the third value overwrites the second
        M4 m;
        NumStream ns(e->get_text());
        m(0, 0) = ns.next_double();
        m(1, 1) = ns.next_double();
        m(1, 1) = ns.next_double();
        node.transform = mul_ref(m, node.transform);
      }
    }
    M4 save = transform;
    node.transform = mul_ref(transform, node.transform);
    transform = node.transform;
    // children are emitted before their parent (:343-347,431)
    for (XmlElement *c = get_element(xml, "node"); c; c = c->next_sibling("node")) parse_node(c);
    transform = save;

    XmlElement *e_camera = get_element(xml, "instance_camera");
    XmlElement *e_light = get_element(xml, "instance_light");
    XmlElement *e_geometry = get_element(xml, "instance_geometry");
    if (e_camera) {
      node.type = INST_CAMERA;
      node.camera = std::make_shared<CameraInfo>();
      parse_camera(e_camera, *node.camera);
    } else if (e_light) {
      node.type = INST_LIGHT;
      node.light = std::make_shared<LightInfo>();
      parse_light(e_light, *node.light);
    } else if (e_geometry) {
      if (get_element(e_geometry, "mesh")) {
        node.type = INST_POLYMESH;
        node.polymesh = std::make_shared<PolymeshInfo>();
        parse_polymesh(e_geometry, *node.polymesh);
        node.polymesh->has_material = bind_material(xml, node.polymesh->material);
      } else if (get_element(e_geometry, "extra")) {
        node.type = INST_SPHERE;
        node.sphere = std::make_shared<SphereInfo>();
        parse_sphere(e_geometry, *node.sphere);
        node.sphere->has_material = bind_material(xml, node.sphere->material);
      }
    }
    // the reference pushes instance-less nodes too and then dereferences the
    // null instance; such nodes are dropped here instead
    if (node.type >= 0) scene->nodes.push_back(node);
  }

  void load(XmlDocument &doc, SceneInfo *out) {  // collada.cpp:133-227
    XmlElement *root = doc.first_child("COLLADA");
    if (!root) throw LoadError("not a COLLADA file");
    scene = out;
    uri_load(root);
    if (XmlElement *e_asset = get_element(root, "asset")) {
      XmlElement *up_axis = get_element(e_asset, "up_axis");
      if (!up_axis) throw LoadError("no up direction defined in COLLADA file");
      std::string up_dir = text_or_throw(up_axis, "up_axis");
      transform = M4::identity();
      if (up_dir == "X_UP") {
        transform(0, 0) = 0; transform(0, 1) = 1;
        transform(1, 0) = 1; transform(1, 1) = 0;
        transform(2, 2) = -1;
        up = V3(1, 0, 0);
      } else if (up_dir == "Z_UP") {
        transform(1, 1) = 0; transform(1, 2) = 1;
        transform(2, 1) = 1; transform(2, 2) = 0;
        transform(0, 0) = -1;
        up = V3(0, 0, 1);
      } else if (up_dir == "Y_UP") {
        up = V3(0, 1, 0);
      } else {
        throw LoadError("invalid up direction in COLLADA file");
      }
    }
    XmlElement *e_scene = get_element(root, "scene/instance_visual_scene");
    if (!e_scene) throw LoadError("no scene description found in file");
    for (XmlElement *n = get_element(e_scene, "node"); n; n = n->next_sibling("node")) parse_node(n);
  }
};

}  // namespace

int collada_load(const char *filename, SceneInfo *out, std::string *err) {
  XmlDocument doc;
  if (!doc.load_file(filename)) {
    if (err) *err = "XML error: " + doc.error;
    return -1;
  }
  try {
    Loader ld;
    ld.load(doc, out);
  } catch (const std::exception &e) {
    if (err) *err = std::string("COLLADA error: ") + e.what();
    return -2;
  }
  return 0;
}

}  // namespace ptamd
