// config.h -- PhnRec's INI configuration, same file format and schema as the reference.
//
// Format (configz.cpp:102-166): "[section]" lines, "var=value" lines (value = everything
// after '=' up to a '#'), lines starting with '#' and empty lines ignored.  Every variable
// must be in the schema (srec.cpp:34-110, doc/config.txt): an unknown one is a fatal error,
// as with Config::SetCheckUnknownVariables(true) (srec.cpp:242).  Typed values are checked
// with the reference's rules: int "%d", float "%f", bool exactly "true"/"false".
#ifndef PHNREC_HOST_CONFIG_H
#define PHNREC_HOST_CONFIG_H

#include <map>
#include <string>

namespace phnrec {

class Config {
public:
    enum Type { STRING, INT, FLOAT, BOOL };
    enum Status { OK = 0, FILEERR, UNKVAR, BADVAL, INVVAR };

    Config();                                   // schema defaults loaded
    Status Load(const std::string &file, int *err_line);
    bool Has(const std::string &section, const std::string &var) const;
    // Like the reference these abort on a key that is not in the schema.
    const std::string &GetString(const std::string &section, const std::string &var) const;
    int GetInt(const std::string &section, const std::string &var) const;
    float GetFloat(const std::string &section, const std::string &var) const;
    bool GetBool(const std::string &section, const std::string &var) const;
    void SetString(const std::string &section, const std::string &var, const std::string &value);

    // "$C..." -> config dir, "$T..." -> dirs/tmp (SpeechRec::SubstVars, srec.cpp:219-233)
    std::string Subst(const std::string &path, const std::string &config_dir) const;

private:
    std::map<std::pair<std::string, std::string>, std::string> values_;
};

}  // namespace phnrec
#endif
