// config.cpp -- see config.h
#include "config.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace phnrec {

namespace {

struct Var { const char *section, *name; Config::Type type; const char *def; };

// The variables a PhnRec configuration may set, with their defaults (the reference's
// cfg_entry table, srec.cpp:34-110; doc/config.txt documents the same list).
const Var kSchema[] = {
    {"source", "format", Config::STRING, "lin16"},
    {"source", "sample_freq", Config::INT, "8000"},
    {"source", "scale", Config::FLOAT, "1.0f"},
    {"source", "dc_shift", Config::FLOAT, "0.0f"},
    {"source", "noise_level", Config::FLOAT, "0.0f"},
    {"params", "kind", Config::STRING, "fbanks"},
    {"params", "suffix", Config::STRING, "mel"},
    {"melbanks", "nbanks", Config::INT, "15"},
    {"melbanks", "nbanks_full", Config::INT, "-1"},
    {"melbanks", "lower_freq", Config::FLOAT, "0"},
    {"melbanks", "higher_freq", Config::FLOAT, "4000"},
    {"melbanks", "vector_size", Config::INT, "200"},
    {"melbanks", "vector_step", Config::INT, "80"},
    {"melbanks", "preem_coef", Config::FLOAT, "0.0"},
    {"melbanks", "z_mean_source", Config::BOOL, "false"},
    {"plp", "order", Config::INT, "12"},
    {"plp", "compress_fact", Config::FLOAT, "0.3333333"},
    {"plp", "cep_lifter", Config::FLOAT, "22"},
    {"plp", "cep_scale", Config::FLOAT, "10"},
    {"plp", "add_c0", Config::BOOL, "false"},
    {"onlinenorm", "estim_interval", Config::INT, "0"},
    {"onlinenorm", "signal_est_end", Config::BOOL, "false"},
    {"onlinenorm", "file", Config::STRING, "none"},
    {"onlinenorm", "mean_norm", Config::BOOL, "false"},
    {"onlinenorm", "var_norm", Config::BOOL, "false"},
    {"onlinenorm", "scale_to_gvar", Config::BOOL, "false"},
    {"offlinenorm", "sent_mean_norm", Config::BOOL, "false"},
    {"offlinenorm", "sent_var_norm", Config::BOOL, "false"},
    {"offlinenorm", "sent_std_thr", Config::FLOAT, "0.01"},
    {"offlinenorm", "sent_max_norm", Config::BOOL, "false"},
    {"offlinenorm", "sent_chmax_norm", Config::BOOL, "false"},
    {"framenorm", "min_floor", Config::FLOAT, "-9999.9"},
    {"framenorm", "shift", Config::FLOAT, "0"},
    {"posteriors", "system", Config::STRING, "1BT_DCT"},
    {"posteriors", "length", Config::INT, "31"},
    {"posteriors", "add_c0", Config::BOOL, "true"},
    {"posteriors", "hamming", Config::BOOL, "false"},
    {"posteriors", "suffix", Config::STRING, "lop"},
    {"posteriors", "bunch_size", Config::STRING, "1"},
    {"posteriors", "enabled", Config::BOOL, "true"},
    {"posteriors", "softening_func", Config::STRING, "none 0 0 0"},
    {"decoder", "type", Config::STRING, "stkint"},
    {"decoder", "wpenalty", Config::FLOAT, "-2.0"},
    {"decoder", "lm_scale", Config::FLOAT, "1.0"},
    {"decoder", "time_pruning", Config::INT, "40"},
    {"decoder", "mode", Config::STRING, "decode"},
    {"decoder", "softening_func", Config::STRING, "log 0 0 0"},
    {"decoder", "num_states_per_phn", Config::INT, "1"},
    {"dirs", "tmp", Config::STRING, "$C/tmp"},
    {"models", "hmm_defs", Config::STRING, "$T/models"},
    {"models", "nstates", Config::INT, "3"},
    {"models", "gen_from_phn_list", Config::BOOL, "false"},
    {"dicts", "phoneme_list", Config::STRING, ""},
    {"dicts", "lexicon1", Config::STRING, ""},
    {"dicts", "lexicon2", Config::STRING, ""},
    {"dicts", "lexicon1_save_bin", Config::BOOL, "false"},
    {"dicts", "lexicon2_save_bin", Config::BOOL, "false"},
    {"dicts", "keyword_list", Config::STRING, "none"},
    {"dicts", "charset", Config::STRING, "eastevrope"},
    {"networks", "default", Config::STRING, "$C/nets/network"},
    {"networks", "gen_phn_loop", Config::BOOL, "false"},
    {"networks", "gen_kws_net", Config::BOOL, "false"},
    {"networks", "omit_phn", Config::STRING, "oth"},
    {"labels", "suffix", Config::STRING, "rec"},
    {"labels", "remove_path", Config::BOOL, "true"},
    {"kws", "default_thr", Config::FLOAT, "-10.0"},
    {"kws", "thresholds_file", Config::STRING, "none"},
    {"gptransc", "rules", Config::STRING, "none"},
    {"gptransc", "symbols", Config::STRING, "none"},
    {"gptransc", "max_variants", Config::INT, "-1"},
    {"gptransc", "scale_prob", Config::BOOL, "false"},
    {"gptransc", "prob_thr", Config::FLOAT, "-1.0"},
    {"phntransc", "mode", Config::STRING, "lexgpt"},
};

const Var *find(const std::string &s, const std::string &v)
{
    for (const Var &e : kSchema)
        if (s == e.section && v == e.name) return &e;
    return nullptr;
}

}  // namespace

Config::Config()
{
    for (const Var &e : kSchema) values_[{e.section, e.name}] = e.def;
}

Config::Status Config::Load(const std::string &file, int *err_line)
{
    std::ifstream in(file.c_str(), std::ios::binary);
    if (!in) return FILEERR;
    std::string line, section;
    int n = 1;
    while (std::getline(in, line)) {
        size_t e = line.find_first_of("\r\n");
        if (e != std::string::npos) line.erase(e);
        if (err_line) *err_line = n;
        if (line.size() > 1 && line[0] == '[') {
            section = line.substr(1, line.size() - 2);           // cut off the last character (']')
        } else if (line.empty() || line[0] == '#') {
        } else {
            // strtok(buff,"=") / strtok(0,"#"): leading '=' are skipped, value ends at '#'
            size_t b = line.find_first_not_of('=');
            size_t eq = b == std::string::npos ? std::string::npos : line.find('=', b);
            if (eq == std::string::npos) return INVVAR;
            std::string var = line.substr(b, eq - b);
            size_t vb = line.find_first_not_of('#', eq + 1);
            if (vb == std::string::npos) return INVVAR;
            size_t ve = line.find('#', vb);
            std::string value = line.substr(vb, ve == std::string::npos ? std::string::npos : ve - vb);
            const Var *sv = find(section, var);
            if (!sv) return UNKVAR;
            int iv; float fv;
            if (sv->type == INT && sscanf(value.c_str(), "%d", &iv) != 1) return BADVAL;
            if (sv->type == FLOAT && sscanf(value.c_str(), "%f", &fv) != 1) return BADVAL;
            if (sv->type == BOOL && value != "true" && value != "false") return BADVAL;
            values_[{section, var}] = value;
        }
        n++;
    }
    return OK;
}

bool Config::Has(const std::string &s, const std::string &v) const { return values_.count({s, v}) != 0; }

const std::string &Config::GetString(const std::string &s, const std::string &v) const
{
    auto it = values_.find({s, v});
    if (it == values_.end()) {           // the reference asserts here (configz.cpp:204)
        fprintf(stderr, "ERROR: configuration variable [%s] %s is not in the schema\n", s.c_str(), v.c_str());
        abort();
    }
    return it->second;
}

int Config::GetInt(const std::string &s, const std::string &v) const
{
    int val = 0;
    sscanf(GetString(s, v).c_str(), "%d", &val);
    return val;
}

float Config::GetFloat(const std::string &s, const std::string &v) const
{
    float val = 0;
    sscanf(GetString(s, v).c_str(), "%f", &val);
    return val;
}

bool Config::GetBool(const std::string &s, const std::string &v) const { return GetString(s, v) == "true"; }

void Config::SetString(const std::string &s, const std::string &v, const std::string &value) { values_[{s, v}] = value; }

std::string Config::Subst(const std::string &path, const std::string &config_dir) const
{
    if (path.size() > 1 && path[0] == '$' && (path[1] == 'C' || path[1] == 'T'))
        return (path[1] == 'C' ? config_dir : GetString("dirs", "tmp")) + path.substr(2);
    return path;
}

}  // namespace phnrec
