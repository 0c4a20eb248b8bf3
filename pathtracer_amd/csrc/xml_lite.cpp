#include "xml_lite.h"

#include <cstring>
#include <fstream>
#include <sstream>

namespace ptamd {

const char *XmlElement::attribute(const char *key) const {
  for (const auto &kv : attrs)
    if (kv.first == key) return kv.second.c_str();
  return nullptr;
}

XmlElement *XmlElement::first_child(const char *nm) const {
  for (const auto &c : children)
    if (!nm || c->name == nm) return c.get();
  return nullptr;
}

XmlElement *XmlElement::next_sibling(const char *nm) const {
  if (!parent) return nullptr;
  for (size_t i = index_in_parent + 1; i < parent->children.size(); i++) {
    XmlElement *c = parent->children[i].get();
    if (!nm || c->name == nm) return c;
  }
  return nullptr;
}

namespace {

struct Parser {
  const std::string &s;
  size_t p = 0;
  std::string err;
  explicit Parser(const std::string &data) : s(data) {}

  bool starts(const char *lit) const { return s.compare(p, strlen(lit), lit) == 0; }
  void skip_ws() {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
  }
  static bool all_ws(const std::string &t) {
    for (char c : t)
      if (!(c == ' ' || c == '\t' || c == '\n' || c == '\r')) return false;
    return true;
  }
  static std::string decode(const std::string &t) {
    if (t.find('&') == std::string::npos) return t;
    std::string o;
    for (size_t i = 0; i < t.size(); i++) {
      if (t[i] != '&') {
        o += t[i];
        continue;
      }
      size_t e = t.find(';', i);
      if (e == std::string::npos) {
        o += t[i];
        continue;
      }
      std::string ent = t.substr(i + 1, e - i - 1);
      if (ent == "lt") o += '<';
      else if (ent == "gt") o += '>';
      else if (ent == "amp") o += '&';
      else if (ent == "quot") o += '"';
      else if (ent == "apos") o += '\'';
      else if (!ent.empty() && ent[0] == '#') {
        long v = (ent.size() > 1 && ent[1] == 'x') ? strtol(ent.c_str() + 2, nullptr, 16) : strtol(ent.c_str() + 1, nullptr, 10);
        if (v > 0 && v < 128) o += (char)v;
      } else {
        o += t.substr(i, e - i + 1);
      }
      i = e;
    }
    return o;
  }
  bool skip_misc() {  // comments, PIs, doctype
    for (;;) {
      skip_ws();
      if (starts("<!--")) {
        size_t e = s.find("-->", p);
        if (e == std::string::npos) return fail("unterminated comment");
        p = e + 3;
      } else if (starts("<?")) {
        size_t e = s.find("?>", p);
        if (e == std::string::npos) return fail("unterminated processing instruction");
        p = e + 2;
      } else if (starts("<!")) {
        size_t e = s.find('>', p);
        if (e == std::string::npos) return fail("unterminated declaration");
        p = e + 1;
      } else {
        return true;
      }
    }
  }
  bool fail(const char *m) {
    if (err.empty()) {
      std::ostringstream o;
      o << m << " at byte " << p;
      err = o.str();
    }
    return false;
  }
  static bool name_char(char c) {
    return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.';
  }
  bool parse_element(XmlElement *parent) {
    // at '<'
    p++;
    size_t b = p;
    while (p < s.size() && name_char(s[p])) p++;
    if (p == b) return fail("bad element name");
    std::unique_ptr<XmlElement> el(new XmlElement());
    el->name = s.substr(b, p - b);
    el->parent = parent;
    el->index_in_parent = parent->children.size();
    // attributes
    for (;;) {
      skip_ws();
      if (p >= s.size()) return fail("unexpected end in tag");
      if (s[p] == '/') {
        if (p + 1 < s.size() && s[p + 1] == '>') {
          p += 2;
          parent->children.push_back(std::move(el));
          return true;
        }
        return fail("bad '/' in tag");
      }
      if (s[p] == '>') {
        p++;
        break;
      }
      size_t ab = p;
      while (p < s.size() && name_char(s[p])) p++;
      if (p == ab) return fail("bad attribute name");
      std::string key = s.substr(ab, p - ab);
      skip_ws();
      if (p >= s.size() || s[p] != '=') return fail("expected '='");
      p++;
      skip_ws();
      if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) return fail("expected quote");
      char q = s[p++];
      size_t vb = p;
      while (p < s.size() && s[p] != q) p++;
      if (p >= s.size()) return fail("unterminated attribute");
      el->attrs.emplace_back(key, decode(s.substr(vb, p - vb)));
      p++;
    }
    // content
    XmlElement *raw = el.get();
    parent->children.push_back(std::move(el));
    bool first_node = true;
    for (;;) {
      size_t tb = p;
      while (p < s.size() && s[p] != '<') p++;
      if (p >= s.size()) return fail("unexpected end in content");
      if (p > tb) {
        std::string t = s.substr(tb, p - tb);
        if (!all_ws(t)) {
          if (first_node) {
            raw->text = decode(t);
            raw->has_text = true;
          }
          first_node = false;
        }
      }
      if (starts("</")) {
        size_t e = s.find('>', p);
        if (e == std::string::npos) return fail("unterminated close tag");
        p = e + 1;
        return true;
      }
      if (starts("<!--")) {
        size_t e = s.find("-->", p);
        if (e == std::string::npos) return fail("unterminated comment");
        p = e + 3;
        continue;
      }
      if (starts("<![CDATA[")) {
        size_t e = s.find("]]>", p);
        if (e == std::string::npos) return fail("unterminated CDATA");
        if (first_node) {
          raw->text = s.substr(p + 9, e - p - 9);
          raw->has_text = true;
        }
        first_node = false;
        p = e + 3;
        continue;
      }
      if (starts("<?")) {
        size_t e = s.find("?>", p);
        if (e == std::string::npos) return fail("unterminated processing instruction");
        p = e + 2;
        continue;
      }
      first_node = false;
      if (!parse_element(raw)) return false;
    }
  }
};

}  // namespace

bool XmlDocument::parse(const std::string &data) {
  root.reset(new XmlElement());
  Parser ps(data);
  for (;;) {
    if (!ps.skip_misc()) break;
    if (ps.p >= data.size()) break;
    if (data[ps.p] != '<') {
      ps.fail("text outside root element");
      break;
    }
    if (!ps.parse_element(root.get())) break;
  }
  error = ps.err;
  return error.empty() && !root->children.empty();
}

bool XmlDocument::load_file(const char *path) {
  std::ifstream in(path, std::ios::binary);
  if (!in.is_open()) {
    error = std::string("cannot open ") + path;
    return false;
  }
  std::stringstream ss;
  ss << in.rdbuf();
  return parse(ss.str());
}

}  // namespace ptamd
