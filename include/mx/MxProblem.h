// MxProblem<DIM>: the `.mx` input deck -> MxEMSim + MxSolver (src/MxProblem.cpp:201-706,
// src/maxwell.cpp:41-120).  The reference reads the deck with Teuchos::XMLObject; here a small
// XML reader (MxUtil::XML::Node) provides the same queries, and MxUtil::XML::getAttr keeps the
// reference's lookup order: body line `name = value` first, then the tag attribute, then the
// default (src/MxUtil.cpp:266-315).
//
// In scope: 3-D, Grid, BoundaryConditions, PEC and Dielectric blocks (loss tangent included)
// with every shape tag of src/MxProblem.cpp:8-17 (Ellipsoid, Sphere, Cylinder, Torus, Cone, Slab, HalfSpace,
// ShapeSubtract / Intersection / Union / Mirror / Repeat) and the generic transforms, Eigensolver /
// LinearSolver / Preconditioner (phase shifts and loss tangents select the complex solver).  Mu / PML
// blocks are rejected with a message instead of being ignored (a wrong answer would be silent otherwise).
//
// Deviation: the reference runs the solve inside the constructor and leaves solve() empty
// (src/MxProblem.cpp:160-175,694); here the constructor parses and assembles, solve() solves.
#ifndef MX_PROBLEM_H
#define MX_PROBLEM_H

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "MxEMSim.h"
#include "MxSolver.h"
#include "MxTypes.h"
#include "mx_geometry.hpp"
#include "mx_yee.hpp"

namespace MxUtil {

namespace Strings {

inline std::string stripLTSpaces(std::string const& s) {   // src/MxUtil.cpp:163-177
  const size_t a = s.find_first_not_of(" \t\r");
  if (a == std::string::npos) return "";
  const size_t b = s.find_last_not_of(" \t\r");
  return s.substr(a, b - a + 1);
}

inline std::pair<std::string, std::string> splitEquation(std::string const& s) {   // src/MxUtil.cpp:182-197
  const size_t pos = s.find('=');
  if (pos == std::string::npos) return std::make_pair(stripLTSpaces(s), stripLTSpaces(s));
  return std::make_pair(stripLTSpaces(s.substr(0, pos)), stripLTSpaces(s.substr(pos + 1)));
}

// "[a, b, c]" -> items; separators are ',', ' ', '[' and ']' (src/MxUtil.cpp:24,96-127)
inline std::vector<std::string> listItems(std::string const& s) {
  std::vector<std::string> res;
  std::string cur;
  for (char ch : s) {
    const bool spacer = ch == ',' || ch == ' ' || ch == '[' || ch == ']' || ch == '\t';
    if (!spacer) cur.push_back(ch);
    else if (!cur.empty()) { res.push_back(cur); cur.clear(); }
  }
  if (!cur.empty()) res.push_back(cur);
  return res;
}

}  // namespace Strings

namespace XML {

// One element of the deck: tag, tag attributes, the body lines that are its own (not those of
// nested elements, src/MxUtil.cpp:227-264) and the nested elements in document order.
class Node {
 public:
  Node() {}
  bool isEmpty() const { return mTag.empty(); }
  std::string const& getTag() const { return mTag; }
  int numChildren() const { return (int)mChildren.size(); }
  Node const& getChild(int i) const { return mChildren.at((size_t)i); }
  int findFirstChild(std::string const& tag) const {
    for (size_t i = 0; i < mChildren.size(); ++i) if (mChildren[i].mTag == tag) return (int)i;
    return -1;
  }
  std::string getWithDefault(std::string const& name, std::string const& dflt) const {
    auto it = mAttrs.find(name);
    return it == mAttrs.end() ? dflt : it->second;
  }
  std::vector<std::string> const& lines() const { return mLines; }

  void print(std::ostream& os, int indent) const {
    const std::string pad((size_t)indent * 2, ' ');
    os << pad << "<" << mTag;
    for (auto const& a : mAttrs) os << " " << a.first << "=\"" << a.second << "\"";
    os << ">\n";
    for (auto const& l : mLines) os << pad << "  " << l << "\n";
    for (auto const& c : mChildren) c.print(os, indent + 1);
    os << pad << "</" << mTag << ">\n";
  }

  static Node parse(std::string const& text) {
    Parser p(text);
    return p.document();
  }
  static Node parseFile(std::string const& path) {
    std::ifstream in(path.c_str());
    if (!in) throw std::runtime_error("could not open input file '" + path + "'");
    std::stringstream ss;
    ss << in.rdbuf();
    return parse(ss.str());
  }

 private:
  std::string mTag;
  std::map<std::string, std::string> mAttrs;
  std::vector<std::string> mLines;
  std::vector<Node> mChildren;

  class Parser {
   public:
    explicit Parser(std::string const& t) : s(t), i(0), line(1) {}
    Node document() {
      skipMisc();
      if (i >= s.size() || s[i] != '<') fail("expected a root element");
      Node root = element();
      skipMisc();
      if (i < s.size()) fail("text after the root element");
      return root;
    }

   private:
    std::string const& s;
    size_t i;
    int line;

    [[noreturn]] void fail(std::string const& what) const {
      std::ostringstream os;
      os << ".mx parse error (line " << line << "): " << what;
      throw std::runtime_error(os.str());
    }
    void advance(size_t n = 1) {
      for (size_t k = 0; k < n && i < s.size(); ++k, ++i) if (s[i] == '\n') ++line;
    }
    bool starts(const char* lit) const { return s.compare(i, std::char_traits<char>::length(lit), lit) == 0; }
    void skipUntil(const char* lit) {
      const size_t pos = s.find(lit, i);
      if (pos == std::string::npos) fail(std::string("unterminated construct, expected '") + lit + "'");
      advance(pos + std::char_traits<char>::length(lit) - i);
    }
    // whitespace, comments, processing instructions and DOCTYPE between elements
    void skipMisc() {
      for (;;) {
        while (i < s.size() && std::isspace((unsigned char)s[i])) advance();
        if (starts("<!--")) skipUntil("-->");
        else if (starts("<?")) skipUntil("?>");
        else if (starts("<!DOCTYPE")) skipUntil(">");
        else return;
      }
    }
    static bool nameChar(char ch) {
      return std::isalnum((unsigned char)ch) || ch == '_' || ch == '-' || ch == ':' || ch == '.';
    }
    std::string name() {
      const size_t a = i;
      while (i < s.size() && nameChar(s[i])) advance();
      if (i == a) fail("expected a name");
      return s.substr(a, i - a);
    }
    static std::string unescape(std::string const& in) {
      static const char* ent[][2] = {{"&lt;", "<"}, {"&gt;", ">"}, {"&amp;", "&"}, {"&quot;", "\""}, {"&apos;", "'"}};
      std::string out;
      for (size_t k = 0; k < in.size();) {
        bool hit = false;
        if (in[k] == '&')
          for (auto& e : ent) {
            const size_t n = std::char_traits<char>::length(e[0]);
            if (in.compare(k, n, e[0]) == 0) { out += e[1]; k += n; hit = true; break; }
          }
        if (!hit) out.push_back(in[k++]);
      }
      return out;
    }
    void addBody(Node& node, std::string const& chunk) {
      std::istringstream is(unescape(chunk));
      std::string l;
      while (std::getline(is, l)) {
        const std::string t = Strings::stripLTSpaces(l);
        if (!t.empty()) node.mLines.push_back(t);
      }
    }
    Node element() {
      Node node;
      advance();                                   // '<'
      node.mTag = name();
      for (;;) {                                   // tag attributes
        while (i < s.size() && std::isspace((unsigned char)s[i])) advance();
        if (i >= s.size()) fail("unterminated tag <" + node.mTag);
        if (s[i] == '>') { advance(); break; }
        if (starts("/>")) { advance(2); return node; }
        const std::string key = name();
        while (i < s.size() && std::isspace((unsigned char)s[i])) advance();
        if (i >= s.size() || s[i] != '=') fail("attribute '" + key + "' without '='");
        advance();
        while (i < s.size() && std::isspace((unsigned char)s[i])) advance();
        if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("attribute '" + key + "' value must be quoted");
        const char q = s[i];
        advance();
        const size_t a = i;
        while (i < s.size() && s[i] != q) advance();
        if (i >= s.size()) fail("unterminated attribute value");
        node.mAttrs[key] = unescape(s.substr(a, i - a));
        advance();
      }
      std::string chunk;
      for (;;) {                                   // content
        if (i >= s.size()) fail("missing </" + node.mTag + ">");
        if (s[i] != '<') { chunk.push_back(s[i]); advance(); continue; }
        if (starts("<!--")) { skipUntil("-->"); continue; }
        if (starts("<![CDATA[")) {
          advance(9);
          const size_t pos = s.find("]]>", i);
          if (pos == std::string::npos) fail("unterminated CDATA");
          chunk += s.substr(i, pos - i);
          advance(pos + 3 - i);
          continue;
        }
        addBody(node, chunk);
        chunk.clear();
        if (starts("</")) {
          advance(2);
          const std::string close = name();
          if (close != node.mTag) fail("</" + close + "> closes <" + node.mTag + ">");
          while (i < s.size() && std::isspace((unsigned char)s[i])) advance();
          if (i >= s.size() || s[i] != '>') fail("malformed closing tag");
          advance();
          return node;
        }
        node.mChildren.push_back(element());
      }
    }
  };
};

inline std::ostream& operator<<(std::ostream& os, Node const& n) { n.print(os, 0); return os; }

inline std::string getBodyAttr(std::string name, Node const& node) {   // src/MxUtil.cpp:294-310
  for (auto const& l : node.lines()) {
    const std::pair<std::string, std::string> eq = Strings::splitEquation(l);
    if (eq.first == name) return eq.second;
  }
  return "";
}
inline std::string getTagAttr(std::string name, Node const& node) { return node.getWithDefault(name, ""); }

inline std::string getAttr(std::string name, Node const& node, std::string dfault) {   // src/MxUtil.cpp:266-280
  if (node.isEmpty()) return dfault;
  const std::string resB = getBodyAttr(name, node), resT = getTagAttr(name, node);
  if (resB != "") return resB;
  if (resT != "") return resT;
  return dfault;
}

// required attribute: the reference prints the block and exits (src/MxUtil.cpp:283-292)
inline std::string getAttr(std::string name, Node const& node) {
  const std::string s = getAttr(name, node, "");
  if (s == "") {
    std::ostringstream os;
    os << "MxUtil::XML::getAttr(...): attribute, '" << name << "', is required to exist in the following XML block:\n";
    node.print(os, 1);
    throw std::runtime_error(os.str());
  }
  return s;
}

}  // namespace XML
}  // namespace MxUtil

namespace mxdetail {

// MxDimVector::strFill (src/MxDimVector.hpp:274-285): missing trailing entries take `fill`
inline void strFill(std::string const& str, double* out, int dim, double fill) {
  const std::vector<std::string> it = MxUtil::Strings::listItems(str);
  if ((int)it.size() < dim)
    std::cout << "MxDimVector::strFill(...): Warning: not enough data in string, '" << str
              << "'. Padding with " << fill << ".\n";
  for (int i = 0; i < dim; ++i) out[i] = i < (int)it.size() ? atof(it[(size_t)i].c_str()) : fill;
}
inline mx::Vec3 vec3(std::string const& str, double fill = 0.0) {
  double v[3];
  strFill(str, v, 3, fill);
  return mx::Vec3(v[0], v[1], v[2]);
}

}  // namespace mxdetail

template <size_t DIM>
class MxProblem {
 public:
  typedef MxUtil::XML::Node XMLNode;

  // `strict` keeps the reference's periodic-wrap quirk in the E/Psi factors.
  // `theComm` may be null: the deck is then only parsed (getConfig / getParameters), nothing is
  // assembled and solve() is unavailable - used by mxh_problem_parse on machines without a GPU.
  MxProblem(XMLNode const& simXML, RCP<MxComm> theComm, bool strict = true)
      : gridSet(false), eigSet(false), bcSet(false), comm(theComm), pid(theComm.is_null() ? 0 : theComm->myPID()),
        mStrict(strict) {
    static_assert(DIM == 3, "only the 3-D path is built (SURVEY §8)");
    if (pid == 0) std::cout << "Running simulation: " << simXML.getTag() << "\n";
    mName = MxUtil::XML::getAttr("name", simXML, simXML.getTag());
    const std::string pol = MxUtil::XML::getAttr("polarization", simXML);   // src/MxProblem.cpp:117-126
    if (pol != "TE" && pol != "TM")
      throw std::runtime_error("Invalid polarization: '" + pol + "'. Should be 'TE' or 'TM'.");
    pList.set("polarization", pol.c_str());
    pList.set("is complex", false);
    pList.set("has curl null", true);
    for (int i = 0; i < simXML.numChildren(); ++i) preParse(simXML.getChild(i));
    if (!eigSet) getEigensolver(XMLNode());
    if (!gridSet) getGrid(XMLNode());
    if (!bcSet) getBCs(XMLNode());
    cfg.strict = mStrict;
    if (comm.is_null()) return;
    if (comm->numProc() > 1) {
      // one rank per MxGridDomain box (src/MxGrid.cpp:171-193): assemble this rank's box + halo window and form the
      // operator chain on it; MxEMSim / MxMagWaveOp take the owned rows from there (real problems)
      int pg[3];
      if (mxh_partition(cfg.n, comm->numProc(), comm->myPID(), pg, cfg.boxLo, cfg.boxHi) != 0)
        throw std::runtime_error("MxProblem: MxGrid::partition failed");
      if (pList.get("is complex", false)) {
        mBoxAsmC = std::make_shared<mx::YeeAssembler<MxComplex> >(cfg);
        mBoxOpsC = mBoxAsmC->waveOps();
        sim = rcp(new MxEMSim<DIM>(cfg, comm, pList, nullptr, mBoxAsmC));
        sim->setPrebuiltOps(&mBoxOpsC);
        return;
      }
      mBoxAsm = std::make_shared<mx::YeeAssembler<double> >(cfg);
      mBoxOps = mBoxAsm->waveOps();
      sim = rcp(new MxEMSim<DIM>(cfg, comm, pList, mBoxAsm));
      sim->setPrebuiltOps(&mBoxOps);
      return;
    }
    sim = rcp(new MxEMSim<DIM>(cfg, comm, pList));
  }

  // reference main(): `prob3d = new MxProblem<3>(xmlObj, myComm); prob3d->solve();`
  void solve() {
    if (sim.is_null()) throw std::runtime_error("MxProblem::solve: parsed without a communicator");
    if (pList.get("is complex", false)) {       // src/MxProblem.cpp:160-175: MxSolver<DIM, MxComplex> when "is complex"
      csolver = rcp(new MxSolver<DIM, MxComplex>(sim, pList));
      csolver->solve();
    } else {
      solver = rcp(new MxSolver<DIM, double>(sim, pList));
      solver->solve();
    }
  }
  bool isComplex() const { return pList.get("is complex", false); }
  std::vector<MxComplex> const& eigenvalues() const { return isComplex() ? csolver->eigenvalues() : solver->eigenvalues(); }
  std::vector<MxComplex> const& frequencies() const { return isComplex() ? csolver->frequencies() : solver->frequencies(); }
  std::vector<double> const& residuals() const { return isComplex() ? csolver->residuals() : solver->residuals(); }

  // MxSolver::solve writes the eigenvector files like the reference (src/MxSolver.cpp:222-224)
  void setSaveFields(bool on) { pList.set("io : save fields", on); }
  Teuchos::ParameterList const& getParameters() const { return pList; }
  mx::YeeConfig const& getConfig() const { return cfg; }
  std::string const& getName() const { return mName; }
  RCP<MxEMSim<DIM> > getSim() const { return sim; }
  RCP<MxSolver<DIM, double> > getSolver() const { return solver; }
  RCP<MxSolver<DIM, MxComplex> > getComplexSolver() const { return csolver; }

  // format: name, default, type  (src/MxProblem.cpp:20-66)
  static const char* const (*eigSolverParams())[3] {
    static const char* const t[10][3] = {
        {"type", "krylov-schur", "string"}, {"basis", "20", "int"},  {"max restarts", "100", "int"},
        {"invert", "true", "bool"},          {"block size", "1", "int"}, {"nev", "10", "int"},
        {"tol", "1.0e-6", "double"},         {"spectrum", "LM", "string"}, {"output", "2", "int"},
        {"shift", "0.0", "double"}};
    return t;
  }
  static const char* const (*linSolverParams())[3] {
    static const char* const t[3][3] = {{"type", "gmres", "string"}, {"tol", "1.0e-6", "double"}, {"basis", "40", "int"}};
    return t;
  }
  static const char* const (*amgParams())[3] {
    static const char* const t[4][3] = {{"smoother", "Chebyshev", "string"}, {"sweeps", "2", "int"},
                                        {"coarse smoother", "Chebyshev", "string"}, {"max levels", "10", "int"}};
    return t;
  }
  static const char* const (*iluParams())[3] {
    static const char* const t[2][3] = {{"drop tol", "1.e-3", "double"}, {"fill", "10.", "double"}};
    return t;
  }

 private:
  void addParam(XMLNode const& node, std::string const& prefix, const char* const set[3]) {   // src/MxProblem.cpp:82-105
    const std::string name = set[0], dflt = set[1], type = set[2];
    const std::string sval = MxUtil::XML::getAttr(name, node, dflt);
    if (type == "int") pList.set(prefix + name, atoi(sval.c_str()));
    else if (type == "double") pList.set(prefix + name, atof(sval.c_str()));
    else if (type == "bool") pList.set(prefix + name, sval == "true");
    else pList.set(prefix + name, sval.c_str());
  }

  static bool isShapeTag(std::string const& tag) {   // src/MxProblem.cpp:8-17
    static const char* tags[] = {"Ellipsoid", "Sphere", "Cylinder", "Torus", "Cone", "Slab", "HalfSpace",
                                 "ShapeSubtract", "ShapeIntersection", "ShapeUnion", "ShapeMirror", "ShapeRepeat"};
    for (auto t : tags) if (tag == t) return true;
    return false;
  }

  void preParse(XMLNode const& node) {   // src/MxProblem.cpp:225-250
    const std::string tag = node.getTag();
    if (isShapeTag(tag)) getShape(node);
    else if (tag == "Eigensolver") getEigensolver(node);
    else if (tag == "Grid") getGrid(node);
    else if (tag == "PEC") getPEC(node);
    else if (tag == "BoundaryConditions") getBCs(node);
    else if (tag == "Dielectric") getDielectric(node);
    else if (tag == "Mu") getMu(node);
    else if (tag == "PML") getPML(node);
    else if (pid == 0)
      std::cout << "MxProblem<" << DIM << ">::parseNode: Did not understand XML tag: '" << tag << "'. Ignoring.\n";
  }

  // generic shape attributes, in document order (src/MxShape.cpp:30-60; "order matters in .mx file")
  static void transforms(mx::Shape& shape, XMLNode const& node) {
    using MxUtil::XML::getAttr;
    if (getAttr("invert", node, "false") == "true") shape.invert();
    for (int i = 0; i < node.numChildren(); ++i) {
      XMLNode const& child = node.getChild(i);
      const std::string tag = child.getTag();
      if (tag == "Rotate") {                                            // src/MxShape.cpp:63-75
        const mx::Vec3 axis = mxdetail::vec3(getAttr("axis", child));
        const double angle = atof(getAttr("angle", child).c_str());
        const std::string s = getAttr("pivot", child, "");
        shape.rotate(axis, angle, s != "" ? mxdetail::vec3(s) : shape.translation());
      } else if (tag == "Translate") {                                  // src/MxShape.cpp:175-181
        shape.translate(mxdetail::vec3(getAttr("vector", child)));
      } else if (tag == "Reflect") {                                    // src/MxShape.cpp:191-197
        shape.reflect(mxdetail::vec3(getAttr("normal", child)), mxdetail::vec3(getAttr("point in plane", child)));
      } else if (tag == "Scale") {                                      // src/MxShape.cpp:129-136
        shape.scale(mxdetail::vec3(getAttr("magnitudes", child), 1.0), mx::Vec3(0, 0, 0));
      }
    }
  }

  // the shape blocks nested in a composite shape, in document order (Translate/Rotate/... are transforms of the composite)
  std::vector<size_t> childShapes(XMLNode const& node) {
    std::vector<size_t> out;
    for (int i = 0; i < node.numChildren(); ++i)
      if (isShapeTag(node.getChild(i).getTag())) out.push_back(getShape(node.getChild(i)));
    return out;
  }

  size_t getShape(XMLNode const& node) {   // src/MxProblem.cpp:277-384
    using MxUtil::XML::getAttr;
    const std::string tag = node.getTag();
    std::shared_ptr<mx::Shape> shape;
    const mx::Vec3 zero(0, 0, 0);
    if (tag == "Sphere") {                                               // src/MxSphere.hpp:53-63
      getAttr("name", node);
      shape = std::make_shared<mx::Sphere>(atof(getAttr("radius", node, "1").c_str()), zero);
      transforms(*shape, node);
    } else if (tag == "Cylinder") {                                      // src/MxCylinder.hpp:74-84
      getAttr("name", node);
      const double r = atof(getAttr("radius", node).c_str());
      shape = std::make_shared<mx::Cylinder>(r, mxdetail::vec3(getAttr("axis", node)), zero);
      transforms(*shape, node);
    } else if (tag == "Slab") {                                          // src/MxSlab.hpp:62-74
      getAttr("name", node);
      const mx::Vec3 n = mxdetail::vec3(getAttr("normal", node));
      shape = std::make_shared<mx::Slab>(atof(getAttr("thickness", node).c_str()), n, zero);
      transforms(*shape, node);
    } else if (tag == "HalfSpace") {                                     // src/MxHalfSpace.hpp:44-49
      getAttr("name", node);
      shape = std::make_shared<mx::HalfSpace>(zero, mxdetail::vec3(getAttr("normal", node)));
      transforms(*shape, node);
    } else if (tag == "ShapeIntersection") {                             // src/MxProblem.cpp:316-326
      auto tmp = std::make_shared<mx::ShapeIntersection>();
      for (size_t indx : childShapes(node)) tmp->add(shapes[indx]);
      shape = tmp;
      transforms(*shape, node);
    } else if (tag == "Ellipsoid") {                                     // src/MxEllipsoid.hpp:54-62
      getAttr("name", node);
      shape = std::make_shared<mx::Ellipsoid>(mxdetail::vec3(getAttr("axes", node)), zero);
      transforms(*shape, node);
    } else if (tag == "Torus") {                                         // src/MxTorus.hpp:69-82
      getAttr("name", node);
      const double R = atof(getAttr("major radius", node).c_str()), r = atof(getAttr("minor radius", node).c_str());
      shape = std::make_shared<mx::Torus>(R, r, mxdetail::vec3(getAttr("axis", node)), zero);
      transforms(*shape, node);
    } else if (tag == "Cone") {                                          // src/MxCone.hpp:64-73
      getAttr("name", node);
      const double angle = atof(getAttr("angle", node).c_str());
      shape = std::make_shared<mx::Cone>(angle, mxdetail::vec3(getAttr("axis", node)), zero);
      transforms(*shape, node);
    } else if (tag == "ShapeUnion") {                                    // src/MxProblem.cpp:327-336
      auto tmp = std::make_shared<mx::ShapeUnion>();
      for (size_t indx : childShapes(node)) tmp->add(shapes[indx]);
      shape = tmp;
      transforms(*shape, node);
    } else if (tag == "ShapeSubtract") {                                 // src/MxProblem.cpp:297-314: base, then the shape removed
      const std::vector<size_t> kids = childShapes(node);
      if (kids.size() != 2) throw std::runtime_error("'ShapeSubtract' should have exactly 2 shape subobjects.");
      auto tmp = std::make_shared<mx::ShapeSubtract>();
      tmp->setBaseShape(shapes[kids[0]]);
      tmp->subtractShape(shapes[kids[1]]);
      shape = tmp;
      transforms(*shape, node);
    } else if (tag == "ShapeMirror") {                                   // src/MxProblem.cpp:337-353
      const mx::Vec3 n = mxdetail::vec3(getAttr("normal", node)), p = mxdetail::vec3(getAttr("point in plane", node));
      const std::vector<size_t> kids = childShapes(node);
      if (kids.size() != 1) throw std::runtime_error("ShapeMirror XML block must contain one shape block.");
      shape = std::make_shared<mx::ShapeMirror>(shapes[kids[0]], n, p);
      transforms(*shape, node);
    } else if (tag == "ShapeRepeat") {                                   // src/MxProblem.cpp:354-375
      const mx::Vec3 o = mxdetail::vec3(getAttr("origin", node)), dir = mxdetail::vec3(getAttr("direction", node));
      const double step = atof(getAttr("step", node).c_str());
      const int numPos = atoi(getAttr("pos steps", node).c_str()), numNeg = atoi(getAttr("neg steps", node).c_str());
      const std::vector<size_t> kids = childShapes(node);
      if (kids.size() != 1) throw std::runtime_error("ShapeRepeat XML block must contain one shape block.");
      shape = std::make_shared<mx::ShapeRepeat>(shapes[kids[0]], o, dir, step, numPos, numNeg);
      transforms(*shape, node);
    } else {
      throw std::runtime_error("MxProblem<3>::getShape: did not understand shape tag '" + tag + "'.");
    }
    shapes.push_back(shape);
    return shapes.size() - 1;
  }

  void getLinearSolver(XMLNode const& node) {   // src/MxProblem.cpp:386-419
    if (pid == 0) std::cout << "Setting linear solver parameters...\n";
    for (int i = 0; i < 3; ++i) addParam(node, "linear solver : ", linSolverParams()[i]);
    const int res = node.findFirstChild("Preconditioner");
    if (res != -1) {
      XMLNode const& prec = node.getChild(res);
      const std::string type = MxUtil::XML::getAttr("type", prec);
      pList.set("linear solver : prec type", type.c_str());
      if (type == "amg") {
        if (pid == 0) std::cout << "  Found AMG preconditioner\n";
        for (int i = 0; i < 4; ++i) addParam(prec, "linear solver : amg : ", amgParams()[i]);
      } else if (type == "ilu") {
        if (pid == 0) std::cout << "  Found ILU preconditioner\n";
        for (int i = 0; i < 2; ++i) addParam(prec, "linear solver : ilu : ", iluParams()[i]);
      } else if (type == "jacobi") {
        // extension of the reference's grammar (amg | ilu): plain diagonal scaling, no parameters
        if (pid == 0) std::cout << "  Found Jacobi (diagonal scaling) preconditioner\n";
      } else {
        std::ostringstream os;
        os << "Did not understand Preconditioner type: '" << type << "'.\n\nOffending XML block:\n" << prec;
        throw std::runtime_error(os.str());
      }
    }
  }

  void getEigensolver(XMLNode const& node) {   // src/MxProblem.cpp:421-443
    if (pid == 0) std::cout << "Setting eigensolver parameters...\n";
    for (int i = 0; i < 10; ++i) addParam(node, "eigensolver : ", eigSolverParams()[i]);
    if (pList.get("eigensolver : invert", true)) {
      const int res = node.findFirstChild("LinearSolver");
      if (res != -1) getLinearSolver(node.getChild(res));
      else if (node.isEmpty()) getLinearSolver(XMLNode());   // no <Eigensolver> at all: all defaults
      else throw std::runtime_error("In eigensolver block, invert = true but no child <LinearSolver> block was given.");
    }
    eigSet = true;
  }

  void getGrid(XMLNode const& node) {   // src/MxProblem.cpp:445-459
    using MxUtil::XML::getAttr;
    double n[3], l[3], o[3];
    mxdetail::strFill(getAttr("resolution", node, "[16, 16, 16]"), n, 3, 0.0);
    mxdetail::strFill(getAttr("dimensions", node, "[1.0, 1.0, 1.0]"), l, 3, 0.0);
    mxdetail::strFill(getAttr("origin", node, "[0.0, 0.0, 0.0]"), o, 3, 0.0);
    for (int i = 0; i < 3; ++i) {
      cfg.n[i] = (int)n[i]; cfg.size[i] = l[i]; cfg.origin[i] = o[i];
      if (cfg.n[i] < 1) throw std::runtime_error("<Grid>: resolution must be >= 1 in every direction");
    }
    gridSet = true;
  }

  void getPEC(XMLNode const& node) {   // src/MxProblem.cpp:461-484
    int nShapes = 0, first = -1;
    for (int i = 0; i < node.numChildren(); ++i)
      if (isShapeTag(node.getChild(i).getTag())) { if (first < 0) first = i; ++nShapes; }
    if (node.numChildren() > 1 || nShapes != 1)
      throw std::runtime_error("PEC object should contain exactly 1 Shape object");
    cfg.dmFrac = atof(MxUtil::XML::getAttr("D-M frac", node, "0.0").c_str());
    const size_t indx = getShape(node.getChild(first));
    cfg.pec = shapes[indx];
  }

  void getDielectric(XMLNode const& node) {   // src/MxProblem.cpp:487-527
    using MxUtil::XML::getAttr;
    int nShapes = 0, first = -1;
    for (int i = 0; i < node.numChildren(); ++i)
      if (isShapeTag(node.getChild(i).getTag())) { if (first < 0) first = i; ++nShapes; }
    if (node.numChildren() > 1 || nShapes != 1)
      throw std::runtime_error("Dielectric object should contain exactly 1 Shape object");
    mx::YeeDielectric d;
    d.name = getAttr("name", node, "dielectric" + std::to_string(cfg.dielectrics.size()));
    if (pid == 0) std::cout << "Parsed dielectric object '" << d.name << "'\n";
    d.shape = shapes[getShape(node.getChild(first))];
    double diag[3], off[3];
    mxdetail::strFill(getAttr("eps diag", node, "[1.0, 1.0, 1.0]"), diag, 3, 0.0);
    mxdetail::strFill(getAttr("eps off diag", node, "[0.0, 0.0, 0.0]"), off, 3, 0.0);
    const double losstan = atof(getAttr("loss tangent", node, "0.0").c_str());
    if (losstan != 0) pList.set("is complex", true);                         // src/MxProblem.cpp:507-508
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d.eps[i][j] = 0.0;
    for (int i = 0; i < 3; ++i) { d.eps[i][i] = diag[i]; d.epsIm[i][i] = diag[i] * losstan; }   // :511-513
    d.eps[0][1] = d.eps[1][0] = off[2];
    d.eps[0][2] = d.eps[2][0] = off[1];
    d.eps[1][2] = d.eps[2][1] = off[0];
    cfg.dielectrics.push_back(d);
  }

  void getMu(XMLNode const& node) {   // src/MxProblem.cpp:515-556
    using MxUtil::XML::getAttr;
    int nShapes = 0, first = -1;
    for (int i = 0; i < node.numChildren(); ++i)
      if (isShapeTag(node.getChild(i).getTag())) { if (first < 0) first = i; ++nShapes; }
    if (node.numChildren() > 1 || nShapes != 1)
      throw std::runtime_error("Mu object should contain exactly 1 Shape object");
    mx::YeeMu m;
    m.name = getAttr("name", node, "mu" + std::to_string(cfg.mus.size()));
    if (pid == 0) std::cout << "Parsed mu object '" << m.name << "'\n";
    m.shape = shapes[getShape(node.getChild(first))];
    double diag[3], off[3];
    mxdetail::strFill(getAttr("mu diag", node, "[1.0, 1.0, 1.0]"), diag, 3, 0.0);
    mxdetail::strFill(getAttr("mu off diag", node, "[0.0, 0.0, 0.0]"), off, 3, 0.0);
    const double losstan = atof(getAttr("loss tangent", node, "0.0").c_str());
    if (losstan != 0) pList.set("is complex", true);                        // :536-537
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.mu[i][j] = 0.0;
    for (int i = 0; i < 3; ++i) { m.mu[i][i] = diag[i]; m.muIm[i][i] = diag[i] * losstan; }   // :545-547
    m.mu[0][1] = m.mu[1][0] = off[2];                                        // :549-551
    m.mu[0][2] = m.mu[2][0] = off[1];
    m.mu[1][2] = m.mu[2][1] = off[0];
    cfg.mus.push_back(m);
  }

  void getPML(XMLNode const& node) {   // src/MxProblem.cpp:559-598
    using MxUtil::XML::getAttr;
    if (node.numChildren() > 1) throw std::runtime_error("PML object should contain no more than 1 Shape object");
    if (node.numChildren() == 0) throw std::runtime_error("PML object needs to contain exactly 1 Shape object");
    mx::YeePml m;
    m.name = getAttr("name", node, "PML" + std::to_string(cfg.pmls.size()));
    m.shape = shapes[getShape(node.getChild(0))];
    m.alphaMax = atof(getAttr("max alpha", node, "1.0").c_str());
    m.exponent = atof(getAttr("exponent", node, "2.0").c_str());
    mxdetail::strFill(getAttr("profile begin", node), m.begin, 3, 0.0);       // required attributes
    mxdetail::strFill(getAttr("profile end", node), m.end, 3, 0.0);
    cfg.pmls.push_back(m);
    pList.set("is complex", true);                                            // :596-597
  }

  void getBCs(XMLNode const& node) {   // src/MxProblem.cpp:604-645
    using MxUtil::XML::getAttr;
    std::vector<std::string> lo = MxUtil::Strings::listItems(getAttr("lower bcs", node, "[periodic, periodic, periodic]"));
    std::vector<std::string> up = MxUtil::Strings::listItems(getAttr("upper bcs", node, "[periodic, periodic, periodic]"));
    lo.resize(3, ""); up.resize(3, "");
    double ph[3];
    mxdetail::strFill(getAttr("phase shifts", node, "[0.0, 0.0, 0.0]"), ph, 3, 0.0);
    for (int i = 0; i < 3; ++i) {
      if (lo[i] != "periodic" && lo[i] != "pec" && lo[i] != "pmc")
        throw std::runtime_error("Invalid lower boundary condition: '" + lo[i] + "'.");
      if (up[i] != "periodic" && up[i] != "pec" && up[i] != "pmc")
        throw std::runtime_error("Invalid upper boundary condition: '" + up[i] + "'.");
      cfg.lower[i] = mx::parseBc(lo[i].c_str());
      cfg.upper[i] = mx::parseBc(up[i].c_str());
      cfg.phaseShifts[i] = ph[i];
      if (ph[i] != 0) pList.set("is complex", true);
    }
    bcSet = true;
  }

  mx::YeeConfig cfg;
  Teuchos::ParameterList pList;
  std::vector<std::shared_ptr<mx::Shape> > shapes;
  std::shared_ptr<mx::YeeAssembler<double> > mBoxAsm;      // decomposed grid: this rank's window assembler and
  typename MxEMSim<DIM>::OpCache mBoxOps;                  // its operator chain (outlive sim / solver: declared first)
  std::shared_ptr<mx::YeeAssembler<MxComplex> > mBoxAsmC;  // the same for complex decks
  typename MxEMSim<DIM>::OpCacheC mBoxOpsC;
  RCP<MxEMSim<DIM> > sim;
  RCP<MxSolver<DIM, double> > solver;
  RCP<MxSolver<DIM, MxComplex> > csolver;
  bool gridSet, eigSet, bcSet;
  RCP<MxComm> comm;
  int pid;
  bool mStrict;
  std::string mName;
};

#endif
