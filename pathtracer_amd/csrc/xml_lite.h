// Minimal XML DOM reader: elements, attributes, character data. Enough for
// COLLADA 1.4 documents; no DTDs, namespaces are kept as part of the name.
// Navigation helpers mirror the few tinyxml2 calls the reference's loader uses
// (FirstChildElement / NextSiblingElement / Attribute / GetText).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ptamd {

struct XmlElement {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlElement>> children;
  // character data that precedes the first child element (what tinyxml2's
  // GetText() returns); has_text is false when the first child is an element
  std::string text;
  bool has_text = false;
  XmlElement *parent = nullptr;
  size_t index_in_parent = 0;

  const char *attribute(const char *key) const;
  XmlElement *first_child(const char *name = nullptr) const;
  XmlElement *next_sibling(const char *name = nullptr) const;
  const char *get_text() const { return has_text ? text.c_str() : nullptr; }
};

struct XmlDocument {
  std::unique_ptr<XmlElement> root;  // synthetic root holding top-level elements
  std::string error;
  bool load_file(const char *path);
  bool parse(const std::string &data);
  XmlElement *first_child(const char *name) const { return root ? root->first_child(name) : nullptr; }
};

}  // namespace ptamd
