#include "kaldi_io.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rs {

void Fail(const std::string &msg) { throw Error(msg); }

std::string ReadFileBytes(const std::string &path) {
  std::ifstream is(path, std::ios::binary);
  if (!is.good()) Fail("cannot open file: " + path);
  std::ostringstream ss;
  ss << is.rdbuf();
  return ss.str();
}

KaldiReader::KaldiReader(const std::string &path) : KaldiReader(ReadFileBytes(path), path) {}

KaldiReader::KaldiReader(std::string bytes, const std::string &name) : buf_(std::move(bytes)), name_(name) {
  if (buf_.size() >= 2 && buf_[0] == '\0' && buf_[1] == 'B') {
    binary_ = true;
    pos_ = 2;
  }
}

int KaldiReader::Get() {
  if (pos_ >= buf_.size()) Fail(name_ + ": unexpected end of file");
  return (unsigned char)buf_[pos_++];
}

void KaldiReader::SkipSpace() {
  while (pos_ < buf_.size() && std::isspace((unsigned char)buf_[pos_])) pos_++;
}

bool KaldiReader::AtEnd() {
  if (!binary_) SkipSpace();
  return pos_ >= buf_.size();
}

int KaldiReader::PeekChar() {
  if (!binary_) SkipSpace();
  return pos_ < buf_.size() ? (unsigned char)buf_[pos_] : -1;
}

std::string KaldiReader::ReadToken() {
  if (!binary_) SkipSpace();
  size_t b = pos_;
  while (pos_ < buf_.size() && !std::isspace((unsigned char)buf_[pos_])) pos_++;
  if (pos_ == b) Fail(name_ + ": expected a token at offset " + std::to_string(b));
  std::string tok = buf_.substr(b, pos_ - b);
  if (pos_ < buf_.size()) pos_++;  // consume exactly one separator (binary data may follow)
  return tok;
}

std::string KaldiReader::PeekToken() {
  size_t save = pos_;
  if (!binary_) SkipSpace();
  if (pos_ >= buf_.size()) { pos_ = save; return ""; }
  size_t b = pos_, e = pos_;
  while (e < buf_.size() && !std::isspace((unsigned char)buf_[e])) e++;
  pos_ = save;
  return buf_.substr(b, e - b);
}

int KaldiReader::PeekTokenChar() {
  std::string t = PeekToken();
  if (t.empty()) return -1;
  if (t[0] == '<' && t.size() > 1) return (unsigned char)t[1];
  return (unsigned char)t[0];
}

void KaldiReader::ExpectToken(const char *tok) {
  size_t at = pos_;
  std::string t = ReadToken();
  if (t != tok) Fail(name_ + ": expected token " + tok + ", got " + t.substr(0, 40) + " at offset " + std::to_string(at));
}

void KaldiReader::ExpectOneOrTwoTokens(const char *a, const char *b) {
  std::string t = ReadToken();
  if (t == a) ExpectToken(b);
  else if (t != b) Fail(name_ + ": expected token " + a + " or " + b + ", got " + t.substr(0, 40));
}

template <typename T>
void KaldiReader::ReadRaw(T *dst, size_t n) {
  size_t bytes = n * sizeof(T);
  if (pos_ + bytes > buf_.size()) Fail(name_ + ": truncated file");
  std::memcpy(dst, buf_.data() + pos_, bytes);
  pos_ += bytes;
}

double KaldiReader::ReadTextNumber() {
  SkipSpace();
  const char *b = buf_.c_str() + pos_;
  char *e = nullptr;
  double v = std::strtod(b, &e);
  if (e == b) Fail(name_ + ": expected a number at offset " + std::to_string(pos_) + " near '" + buf_.substr(pos_, 16) + "'");
  pos_ += e - b;
  return v;
}

int32_t KaldiReader::ReadInt32() {
  if (binary_) {
    int sz = (signed char)Get();
    if (sz != 4) Fail(name_ + ": expected 4-byte integer, size byte " + std::to_string(sz) + " at offset " + std::to_string(pos_));
    int32_t v;
    ReadRaw(&v, 1);
    return v;
  }
  double v = ReadTextNumber();
  return (int32_t)v;
}

double KaldiReader::ReadDouble() {
  if (binary_) {
    int sz = (signed char)Get();
    if (sz == 4) { float f; ReadRaw(&f, 1); return f; }
    if (sz == 8) { double d; ReadRaw(&d, 1); return d; }
    Fail(name_ + ": expected float/double, size byte " + std::to_string(sz) + " at offset " + std::to_string(pos_));
  }
  return ReadTextNumber();
}

float KaldiReader::ReadFloat() { return (float)ReadDouble(); }

void KaldiReader::ReadBasicAny(double *as_float, int64_t *as_int) {
  if (binary_) {
    int sz = (signed char)Get();
    if (sz == 4) {
      uint32_t raw;
      ReadRaw(&raw, 1);
      float f;
      int32_t i;
      std::memcpy(&f, &raw, 4);
      std::memcpy(&i, &raw, 4);
      *as_float = f;
      *as_int = i;
      return;
    }
    if (sz == 8) {
      uint64_t raw;
      ReadRaw(&raw, 1);
      double d;
      int64_t i;
      std::memcpy(&d, &raw, 8);
      std::memcpy(&i, &raw, 8);
      *as_float = d;
      *as_int = i;
      return;
    }
    Fail(name_ + ": expected a basic value, size byte " + std::to_string(sz) + " at offset " + std::to_string(pos_));
  }
  double v = ReadTextNumber();
  *as_float = v;
  *as_int = (int64_t)std::llround(v);
}

bool KaldiReader::ReadBool() {
  if (!binary_) SkipSpace();
  int c = Get();
  if (c != 'T' && c != 'F') Fail(name_ + ": expected T/F boolean at offset " + std::to_string(pos_));
  if (!binary_ && pos_ < buf_.size() && std::isspace((unsigned char)buf_[pos_])) pos_++;
  return c == 'T';
}

void KaldiReader::ReadIntVector(std::vector<int32_t> *v) {
  v->clear();
  if (binary_) {
    int sz = (signed char)Get();
    if (sz != 4) Fail(name_ + ": integer vector with element size " + std::to_string(sz));
    int32_t n;
    ReadRaw(&n, 1);
    if (n < 0) Fail(name_ + ": negative vector size");
    v->resize(n);
    if (n) ReadRaw(v->data(), n);
    return;
  }
  SkipSpace();
  if (Get() != '[') Fail(name_ + ": expected '[' for integer vector");
  while (true) {
    SkipSpace();
    if (PeekChar() == ']') { pos_++; break; }
    v->push_back((int32_t)ReadTextNumber());
  }
}

void KaldiReader::ReadTextMatrix(std::vector<double> *vals, int *rows, int *cols) {
  // matrix/kaldi-matrix.cc Matrix::Read text branch: '[' numbers, rows end at '\n' or ';', ']' ends.
  SkipSpace();
  if (Get() != '[') Fail(name_ + ": expected '[' at offset " + std::to_string(pos_));
  vals->clear();
  *rows = 0;
  *cols = 0;
  int cur = 0;
  auto end_row = [&]() {
    if (cur == 0) return;
    if (*cols == 0) *cols = cur;
    else if (*cols != cur) Fail(name_ + ": ragged text matrix");
    (*rows)++;
    cur = 0;
  };
  while (true) {
    if (pos_ >= buf_.size()) Fail(name_ + ": unterminated matrix");
    char c = buf_[pos_];
    if (c == ']') { pos_++; end_row(); break; }
    if (c == '\n' || c == ';') { pos_++; end_row(); continue; }
    if (std::isspace((unsigned char)c)) { pos_++; continue; }
    const char *b = buf_.c_str() + pos_;
    char *e = nullptr;
    double v = std::strtod(b, &e);
    if (e == b) Fail(name_ + ": bad number in text matrix near '" + buf_.substr(pos_, 16) + "'");
    pos_ += e - b;
    vals->push_back(v);
    cur++;
  }
}

void KaldiReader::ReadAnyMatrix(std::vector<double> *dv, std::vector<float> *fv, int *rows, int *cols, bool want_double) {
  if (!binary_) {
    std::vector<double> vals;
    ReadTextMatrix(&vals, rows, cols);
    if (want_double) *dv = std::move(vals);
    else fv->assign(vals.begin(), vals.end());
    return;
  }
  std::string tok = ReadToken();
  if (tok != "FM" && tok != "DM") {
    if (tok == "CM" || tok == "CM2" || tok == "CM3") Fail(name_ + ": compressed matrices (CM) are not supported on this path");
    Fail(name_ + ": expected FM/DM matrix header, got " + tok.substr(0, 16));
  }
  *rows = ReadInt32();
  *cols = ReadInt32();
  size_t n = (size_t)*rows * *cols;
  if (tok == "FM") {
    std::vector<float> tmp(n);
    if (n) ReadRaw(tmp.data(), n);
    if (want_double) dv->assign(tmp.begin(), tmp.end());
    else *fv = std::move(tmp);
  } else {
    std::vector<double> tmp(n);
    if (n) ReadRaw(tmp.data(), n);
    if (want_double) *dv = std::move(tmp);
    else fv->assign(tmp.begin(), tmp.end());
  }
}

void KaldiReader::ReadMatrix(MatF *m) {
  std::vector<double> dv;
  ReadAnyMatrix(&dv, &m->d, &m->rows, &m->cols, false);
}

void KaldiReader::ReadMatrixD(MatD *m) {
  std::vector<float> fv;
  ReadAnyMatrix(&m->d, &fv, &m->rows, &m->cols, true);
}

void KaldiReader::ReadVectorD(std::vector<double> *v) {
  if (!binary_) {
    int r, c;
    ReadTextMatrix(v, &r, &c);
    if (r > 1) Fail(name_ + ": expected a vector, got a matrix");
    return;
  }
  std::string tok = ReadToken();
  if (tok != "FV" && tok != "DV") Fail(name_ + ": expected FV/DV vector header, got " + tok.substr(0, 16));
  int32_t n = ReadInt32();
  if (tok == "FV") {
    std::vector<float> tmp(n);
    if (n) ReadRaw(tmp.data(), n);
    v->assign(tmp.begin(), tmp.end());
  } else {
    v->resize(n);
    if (n) ReadRaw(v->data(), n);
  }
}

void KaldiReader::ReadVector(std::vector<float> *v) {
  std::vector<double> d;
  ReadVectorD(&d);
  v->assign(d.begin(), d.end());
}

void KaldiReader::ReadSpMatrixD(int *dim, std::vector<double> *packed) {
  if (!binary_) {
    // text: rows of the lower triangle, one per line (matrix/packed-matrix.cc)
    SkipSpace();
    if (Get() != '[') Fail(name_ + ": expected '[' for packed matrix");
    packed->clear();
    while (true) {
      SkipSpace();
      if (PeekChar() == ']') { pos_++; break; }
      packed->push_back(ReadTextNumber());
    }
    int n = (int)std::llround((std::sqrt(8.0 * packed->size() + 1.0) - 1.0) / 2.0);
    if ((size_t)n * (n + 1) / 2 != packed->size()) Fail(name_ + ": bad packed matrix size");
    *dim = n;
    return;
  }
  std::string tok = ReadToken();
  if (tok != "FP" && tok != "DP") Fail(name_ + ": expected FP/DP packed matrix header, got " + tok.substr(0, 16));
  int32_t n = ReadInt32();
  *dim = n;
  size_t cnt = (size_t)n * (n + 1) / 2;
  if (tok == "FP") {
    std::vector<float> tmp(cnt);
    if (cnt) ReadRaw(tmp.data(), cnt);
    packed->assign(tmp.begin(), tmp.end());
  } else {
    packed->resize(cnt);
    if (cnt) ReadRaw(packed->data(), cnt);
  }
}

std::string KaldiReader::ReadLine() {
  size_t b = pos_;
  while (pos_ < buf_.size() && buf_[pos_] != '\n') pos_++;
  std::string line = buf_.substr(b, pos_ - b);
  if (pos_ < buf_.size()) pos_++;
  if (!line.empty() && line.back() == '\r') line.pop_back();
  return line;
}

static std::string Trim(const std::string &s) {
  size_t b = s.find_first_not_of(" \t\r\n");
  if (b == std::string::npos) return "";
  size_t e = s.find_last_not_of(" \t\r\n");
  return s.substr(b, e - b + 1);
}

std::vector<std::pair<std::string, std::string>> ReadConfigFile(const std::string &path) {
  std::ifstream is(path);
  if (!is.good()) Fail("Cannot open config file: " + path);
  std::vector<std::pair<std::string, std::string>> out;
  std::string line;
  int ln = 0;
  while (std::getline(is, line)) {
    ln++;
    size_t p = line.find('#');
    if (p != std::string::npos) line.erase(p);
    line = Trim(line);
    if (line.empty()) continue;
    if (line.compare(0, 2, "--") != 0)
      Fail("Reading config file " + path + ": line " + std::to_string(ln) + " does not look like --x=y");
    std::string key, value;
    size_t eq = line.find('=');
    if (eq == std::string::npos) { key = line.substr(2); value = "true"; }
    else { key = line.substr(2, eq - 2); value = Trim(line.substr(eq + 1)); }
    for (auto &c : key) { if (c == '_') c = '-'; c = (char)std::tolower((unsigned char)c); }
    out.emplace_back(key, value);
  }
  return out;
}

}  // namespace rs
