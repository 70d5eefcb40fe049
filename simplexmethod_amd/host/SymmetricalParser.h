// SymmetricalParser.h — text format of /root/reference/input_symmetric.txt
// (same interface as /root/reference/src/SymmetricalParser.h:14-55).
//
//   maximize | max | minimize | min
//   objective[:]            then one or more lines of coefficients
//   constraints[:] | subject to[:]   then one row per line, LAST number = right-hand side
//   '#' starts a comment; blank lines and CR/LF line endings are ignored.
#pragma once

#include <istream>
#include <memory>
#include <string>

#include "Symmetrical.h"

class SymmetricalParser {
public:
    std::unique_ptr<Symmetrical> ParseFromFile(const std::string& filename);
    std::unique_ptr<Symmetrical> ParseFromString(const std::string& content);
    std::string GetLastError() const { return lastError_; }   // non-empty after a nullptr result

private:
    std::unique_ptr<Symmetrical> ParseFromStream(std::istream& stream);
    std::string lastError_;
};
