// utils/InputFile.h (reference: src/utils/InputFile.h:97-185) — the parameter-file reader UAMMD programs use
// (examples/misc/benchmark.cu:185-198, examples/basic_concepts/9-reading_parameters.cu:72-94).
//
// File format: one option per line, `name arg1 arg2 ...`; blank lines and lines whose first non-blank character is '#' are skipped; a
// line `shell <command>` runs the command when the file is read.  getOption(name, Required | Optional) hands back an input stream placed
// on the option's arguments: `in.getOption("dt", InputFile::Required) >> dt;`.  A missing Required option logs and throws
// std::runtime_error; a missing Optional one returns a stream in the failed state, so `if (!in.getOption("flag"))` tests presence.
// The first occurrence of a name wins.  (As in the reference the returned stream is one shared object: use it before the next call.)
#ifndef UAMMD_MI355X_UTILS_INPUTFILE_H
#define UAMMD_MI355X_UTILS_INPUTFILE_H

#include "../uammd.h"

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace uammd {

class InputFile {
  std::string fileName;
  std::vector<std::pair<std::string, std::string>> options;  // name -> the rest of its line

  void takeLine(const std::string &line) {
    std::istringstream words(line);
    std::string name;
    if (!(words >> name) || name[0] == '#') return;
    std::string rest;
    std::getline(words, rest);
    if (name == "shell") {
      const int rc = std::system(rest.c_str());
      if (rc != 0) System::log<System::ERROR>("[InputFile] Shell command execution failed with code %d: %s", rc, rest.c_str());
      return;
    }
    options.emplace_back(name, rest);
  }

public:
  enum OptionType { Required, Optional };

  InputFile(std::string name, shared_ptr<System> sys = nullptr) : fileName(std::move(name)) {
    (void)sys;
    std::ifstream in(fileName);
    if (!in) {
      System::log<System::ERROR>("[InputFile] ERROR: Could not open file %s!.", fileName.c_str());
      return;  // (the reference carries on with no options too: the first Required one then throws)
    }
    for (std::string line; std::getline(in, line);) takeLine(line);
  }

  std::istringstream &getOption(std::string op, OptionType type = OptionType::Optional) {
    static std::istringstream answer;
    answer.str(std::string());
    answer.clear();
    for (const auto &o : options)
      if (o.first == op) {
        answer.str(o.second);
        return answer;
      }
    if (type == Required) {
      System::log<System::ERROR>("[InputFile] Option %s not found in %s!", op.c_str(), fileName.c_str());
      throw std::runtime_error("Required option not found in file " + fileName);
    }
    answer.setstate(std::ios::failbit);
    if (op == "shell") System::log<System::ERROR>("[InputFile] Ignoring use of the reserved \"shell\" option");
    return answer;
  }
};

}  // namespace uammd
#endif
