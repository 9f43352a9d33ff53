// htk.h -- HTK parameter files, the reference's `-t par` / `-t post` dump format
// (matrix.h:75-82, 2506-2573): big-endian int32 nSamples, int32 sampPeriod (100000),
// int16 sampSize (4*cols), int16 paramKind (6), then big-endian float32 rows.
#ifndef PHNREC_HOST_HTK_H
#define PHNREC_HOST_HTK_H

#include <string>
#include <vector>

namespace phnrec {

bool SaveHTK(const std::string &path, const float *data, int rows, int cols);
// `be_words` already holds big-endian 32-bit words (the device wrote them that way)
bool SaveHTKRaw(const std::string &path, const void *be_words, int rows, int cols);
// the same dump written in pieces (a file longer than one launch): header for `rows` rows, then row ranges as they come
FILE *BeginHTKRaw(const std::string &path, int rows, int cols);
bool AppendHTKRaw(FILE *f, const void *be_words, int rows, int cols);
bool LoadHTK(const std::string &path, std::vector<float> &data, int *rows, int *cols);

}  // namespace phnrec
#endif
