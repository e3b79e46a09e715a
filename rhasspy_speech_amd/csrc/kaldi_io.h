// Kaldi object-file reader (binary and text modes) over an in-memory buffer.
// Replaces, for the files on the hot path only, kaldi/src/base/io-funcs{.cc,-inl.h}
// (ReadToken/ExpectToken/ReadBasicType/ReadIntegerVector) and the Read() methods of
// matrix/kaldi-matrix.cc, kaldi-vector.cc, sp-matrix (tokens FM/DM/FV/DV/FP/DP).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace rs {

struct Error : std::runtime_error {
  explicit Error(const std::string &m) : std::runtime_error(m) {}
};
[[noreturn]] void Fail(const std::string &msg);

template <typename T>
struct Mat {
  int rows = 0, cols = 0;
  std::vector<T> d;
  T &operator()(int r, int c) { return d[(size_t)r * cols + c]; }
  const T &operator()(int r, int c) const { return d[(size_t)r * cols + c]; }
  void Resize(int r, int c) { rows = r; cols = c; d.assign((size_t)r * c, T(0)); }
};
using MatF = Mat<float>;
using MatD = Mat<double>;

std::string ReadFileBytes(const std::string &path);

class KaldiReader {
 public:
  // Reads the whole file; detects the "\0B" binary header (base/io-funcs-inl.h:291-320).
  explicit KaldiReader(const std::string &path);
  KaldiReader(std::string bytes, const std::string &name);

  bool binary() const { return binary_; }
  const std::string &name() const { return name_; }
  bool AtEnd();
  bool RawEof() const { return pos_ >= buf_.size(); }   // no whitespace skipping

  // Tokens are ASCII runs terminated by whitespace (base/io-funcs.cc:134-168).
  std::string ReadToken();
  // First character of the next token after '<' (like PeekToken: used to test optional fields), -1 at EOF.
  int PeekTokenChar();
  std::string PeekToken();
  void ExpectToken(const char *tok);
  // Accepts "<A> <B>" or just "<B>" (ExpectOneOrTwoTokens).
  void ExpectOneOrTwoTokens(const char *a, const char *b);

  int32_t ReadInt32();
  float ReadFloat();     // accepts a 4- or 8-byte float in binary mode
  double ReadDouble();   // idem
  bool ReadBool();
  // A basic value whose type (int32/int64 vs float/double) is only known to the consumer: returns both readings.
  void ReadBasicAny(double *as_float, int64_t *as_int);
  void ReadIntVector(std::vector<int32_t> *v);
  void ReadVector(std::vector<float> *v);    // FV or DV (converted)
  void ReadVectorD(std::vector<double> *v);  // FV or DV
  void ReadMatrix(MatF *m);                  // FM or DM
  void ReadMatrixD(MatD *m);
  void ReadSpMatrixD(int *dim, std::vector<double> *packed);  // FP/DP packed lower triangle

  // Raw text access for the nnet3 config section (terminated by an empty line).
  std::string ReadLine();

  // In text mode: true if the next non-space character starts a number/bracket rather than a '<' token.
  int PeekChar();
  size_t pos() const { return pos_; }

 private:
  void SkipSpace();
  int Get();
  template <typename T> void ReadRaw(T *dst, size_t n);
  double ReadTextNumber();
  void ReadTextMatrix(std::vector<double> *vals, int *rows, int *cols);
  void ReadAnyMatrix(std::vector<double> *dv, std::vector<float> *fv, int *rows, int *cols, bool want_double);

  std::string buf_, name_;
  size_t pos_ = 0;
  bool binary_ = false;
};

// Kaldi "--name=value" config files (util/parse-options.cc:459-496): one option per line,
// '#' comments, blank lines ignored.  Returns (name, value) in file order.
std::vector<std::pair<std::string, std::string>> ReadConfigFile(const std::string &path);

}  // namespace rs
