// Protobuf text-format parser for the schema subset in caffe/proto/caffe_param.hpp.
// Behavioural model: google::protobuf::TextFormat as used by ReadProtoFromTextFile
// (src/caffe/util/io.cpp:34-44); feature list from the 23 deploy prototxts.
#include "caffe/proto/caffe_param.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>

namespace caffe {

int& logging::verbosity() {
  static int v = [] { const char* e = std::getenv("MSCNN_LOG_INFO"); return e && *e && *e != '0' ? 1 : 0; }();
  return v;
}

const TextMessage& TextMessage::Empty() {
  static const TextMessage e;
  return e;
}

int TextMessage::count(const std::string& name) const {
  int c = 0;
  for (const auto& f : fields) c += (f.name == name);
  return c;
}

static const TextField* find_field(const TextMessage& m, const std::string& name, int i) {
  for (const auto& f : m.fields)
    if (f.name == name && i-- == 0) return &f;
  return nullptr;
}

const std::string& TextMessage::str(const std::string& name, int i) const {
  const TextField* f = find_field(*this, name, i);
  CHECK(f != nullptr) << "field '" << name << "'[" << i << "] not present";
  CHECK(!f->msg) << "field '" << name << "' is a message, not a scalar";
  return f->scalar;
}

const TextMessage& TextMessage::sub(const std::string& name, int i) const {
  const TextField* f = find_field(*this, name, i);
  if (!f) return Empty();
  CHECK(f->msg != nullptr) << "field '" << name << "' is a scalar, not a message";
  return *f->msg;
}

TextMessage* TextMessage::mutable_sub(const std::string& name, int i) {
  for (auto& f : fields)
    if (f.name == name && i-- == 0) return f.msg.get();
  return nullptr;
}

double TextMessage::num(const std::string& name, int i, double dflt) const {
  const TextField* f = find_field(*this, name, i);
  if (!f) return dflt;
  CHECK(!f->msg) << "field '" << name << "' is a message";
  char* end = nullptr;
  const double v = std::strtod(f->scalar.c_str(), &end);
  CHECK(end && *end == '\0' && end != f->scalar.c_str()) << "field '" << name << "': '" << f->scalar << "' is not a number";
  return v;
}

bool TextMessage::boolean(const std::string& name, bool dflt) const {
  const TextField* f = find_field(*this, name, 0);
  if (!f) return dflt;
  const std::string& s = f->scalar;
  if (s == "true" || s == "True" || s == "t" || s == "1") return true;
  if (s == "false" || s == "False" || s == "f" || s == "0") return false;
  LOG(FATAL) << "field '" << name << "': '" << s << "' is not a bool";
}

void TextMessage::add_scalar(const std::string& name, const std::string& v) { fields.push_back(TextField{name, v, nullptr}); }

TextMessage* TextMessage::add_message(const std::string& name) {
  fields.push_back(TextField{name, "", std::make_shared<TextMessage>()});
  return fields.back().msg.get();
}

void TextMessage::clear(const std::string& name) {
  std::vector<TextField> keep;
  for (auto& f : fields)
    if (f.name != name) keep.push_back(f);
  fields.swap(keep);
}

void TextMessage::set_scalar(const std::string& name, int i, const std::string& v) {
  for (auto& f : fields)
    if (f.name == name && i-- == 0) { f.scalar = v; return; }
  LOG(FATAL) << "set_scalar: field '" << name << "' index out of range";
}

std::string TextMessage::DebugString(int indent) const {
  std::string out, pad(indent, ' ');
  for (const auto& f : fields) {
    if (f.msg) out += pad + f.name + " {\n" + f.msg->DebugString(indent + 2) + pad + "}\n";
    else out += pad + f.name + ": " + f.scalar + "\n";
  }
  return out;
}

namespace {
class Parser {
 public:
  Parser(const std::string& s, const std::string& origin) : s_(s), origin_(origin) {}
  TextMessagePtr ParseTop() {
    TextMessagePtr m = std::make_shared<TextMessage>();
    ParseFields(m.get(), /*until_brace=*/false);
    return m;
  }

 private:
  void Fail(const std::string& why) {
    LOG(FATAL) << origin_ << ":" << line_ << ": prototxt parse error: " << why;
  }
  void SkipWs() {
    while (pos_ < s_.size()) {
      const char c = s_[pos_];
      if (c == '\n') { ++line_; ++pos_; }
      else if (std::isspace((unsigned char)c) || c == ',' || c == ';') ++pos_;
      else if (c == '#') { while (pos_ < s_.size() && s_[pos_] != '\n') ++pos_; }
      else break;
    }
  }
  std::string Ident() {
    size_t b = pos_;
    while (pos_ < s_.size() && (std::isalnum((unsigned char)s_[pos_]) || s_[pos_] == '_' || s_[pos_] == '.')) ++pos_;
    if (b == pos_) Fail(std::string("expected identifier, got '") + (pos_ < s_.size() ? s_[pos_] : '$') + "'");
    return s_.substr(b, pos_ - b);
  }
  std::string Scalar() {
    const char c = s_[pos_];
    if (c == '"' || c == '\'') {
      std::string out;
      // adjacent string literals concatenate, as in TextFormat
      while (pos_ < s_.size() && (s_[pos_] == '"' || s_[pos_] == '\'')) {
        const char q = s_[pos_++];
        while (pos_ < s_.size() && s_[pos_] != q) {
          if (s_[pos_] == '\\' && pos_ + 1 < s_.size()) {
            const char e = s_[pos_ + 1];
            out += (e == 'n' ? '\n' : e == 't' ? '\t' : e);
            pos_ += 2;
          } else {
            if (s_[pos_] == '\n') ++line_;
            out += s_[pos_++];
          }
        }
        if (pos_ >= s_.size()) Fail("unterminated string");
        ++pos_;
        SkipWs();
      }
      return out;
    }
    size_t b = pos_;
    while (pos_ < s_.size() && (std::isalnum((unsigned char)s_[pos_]) || s_[pos_] == '_' || s_[pos_] == '.' ||
                                s_[pos_] == '-' || s_[pos_] == '+'))
      ++pos_;
    if (b == pos_) Fail("expected a value");
    return s_.substr(b, pos_ - b);
  }
  void ParseFields(TextMessage* m, bool until_brace) {
    for (;;) {
      SkipWs();
      if (pos_ >= s_.size()) {
        if (until_brace) Fail("missing '}'");
        return;
      }
      if (s_[pos_] == '}' || s_[pos_] == '>') {
        if (!until_brace) Fail("unexpected '}'");
        ++pos_;
        return;
      }
      const std::string name = Ident();
      SkipWs();
      bool colon = false;
      if (pos_ < s_.size() && s_[pos_] == ':') { colon = true; ++pos_; SkipWs(); }
      if (pos_ >= s_.size()) Fail("unexpected end of input after '" + name + "'");
      if (s_[pos_] == '{' || s_[pos_] == '<') {
        ++pos_;
        ParseFields(m->add_message(name), true);
      } else if (s_[pos_] == '[') {          // short repeated form: f: [1, 2, 3]
        ++pos_;
        for (;;) {
          SkipWs();
          if (pos_ < s_.size() && s_[pos_] == ']') { ++pos_; break; }
          m->add_scalar(name, Scalar());
        }
      } else {
        if (!colon) Fail("expected ':' or '{' after '" + name + "'");
        m->add_scalar(name, Scalar());
      }
    }
  }
  const std::string& s_;
  std::string origin_;
  size_t pos_ = 0;
  int line_ = 1;
};
}  // namespace

TextMessagePtr ParseTextFormat(const std::string& text, const std::string& origin) { return Parser(text, origin).ParseTop(); }

bool ReadFileToString(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) return false;
  out->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  return true;
}

LayerParameter NetParameter::layer(int i) const {
  int k = 0;
  for (const auto& f : m_->fields)
    if (f.name == "layer" && k++ == i) {
      CHECK(f.msg != nullptr) << "layer must be a message";
      return LayerParameter(f.msg);
    }
  LOG(FATAL) << "layer index " << i << " out of range";
}

bool ReadProtoFromTextFile(const std::string& filename, NetParameter* param) {
  std::string text;
  if (!ReadFileToString(filename, &text)) return false;
  *param = NetParameter(ParseTextFormat(text, filename));
  return true;
}

void ReadNetParamsFromTextFileOrDie(const std::string& filename, NetParameter* param) {
  CHECK(ReadProtoFromTextFile(filename, param)) << "Failed to parse NetParameter file: " << filename;
}

NetParameter NetParameterFromString(const std::string& text) { return NetParameter(ParseTextFormat(text)); }

}  // namespace caffe
