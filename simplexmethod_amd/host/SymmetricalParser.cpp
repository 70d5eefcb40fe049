// SymmetricalParser.cpp — see SymmetricalParser.h for the format
// (behaviour restated from /root/reference/src/SymmetricalParser.cpp:28-193).
#include "SymmetricalParser.h"

#include <fstream>
#include <sstream>
#include <vector>

namespace {

std::string strip(const std::string& raw) {
    std::string s = raw.substr(0, raw.find('#'));
    const char* ws = " \t\r\n";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}

std::vector<double> numbers(const std::string& line) {
    std::istringstream in(line);
    std::vector<double> v;
    double x;
    while (in >> x) v.push_back(x);
    return v;
}

}  // namespace

std::unique_ptr<Symmetrical> SymmetricalParser::ParseFromFile(const std::string& filename) {
    std::ifstream f(filename);
    if (!f.is_open()) {
        lastError_ = "cannot open file: " + filename;
        return nullptr;
    }
    return ParseFromStream(f);
}

std::unique_ptr<Symmetrical> SymmetricalParser::ParseFromString(const std::string& content) {
    std::istringstream s(content);
    return ParseFromStream(s);
}

std::unique_ptr<Symmetrical> SymmetricalParser::ParseFromStream(std::istream& stream) {
    enum class Section { None, Objective, Constraints } section = Section::None;
    bool maximize = true;
    std::vector<double> obj, rhs;
    std::vector<std::vector<double>> rows;
    std::string raw;
    while (std::getline(stream, raw)) {
        const std::string line = strip(raw);
        if (line.empty()) continue;
        if (line == "maximize" || line == "max") { maximize = true; continue; }
        if (line == "minimize" || line == "min") { maximize = false; continue; }
        if (line == "objective" || line == "objective:") { section = Section::Objective; continue; }
        if (line == "constraints" || line == "constraints:" || line == "subject to" ||
            line == "subject to:") {
            section = Section::Constraints;
            continue;
        }
        std::vector<double> v = numbers(line);
        if (section == Section::None) {
            lastError_ = "data outside of a section: " + line;
            return nullptr;
        }
        if (section == Section::Objective) {
            obj.insert(obj.end(), v.begin(), v.end());
        } else {
            if (v.size() < 2) {
                lastError_ = "constraint row needs at least one coefficient and a right-hand side: " + line;
                return nullptr;
            }
            rhs.push_back(v.back());
            v.pop_back();
            rows.push_back(v);
        }
    }
    if (obj.empty()) { lastError_ = "no objective given"; return nullptr; }
    if (rows.empty()) { lastError_ = "no constraints given"; return nullptr; }
    const long n = (long)obj.size(), m = (long)rows.size();
    for (const auto& r : rows)
        if ((long)r.size() != n) {
            lastError_ = "constraint row length differs from the objective's";
            return nullptr;
        }
    lpla::MatrixXd A(m, n);
    lpla::VectorXd b(m), c(n);
    for (long i = 0; i < m; ++i) {
        b[i] = rhs[(size_t)i];
        for (long j = 0; j < n; ++j) A(i, j) = rows[(size_t)i][(size_t)j];
    }
    for (long j = 0; j < n; ++j) c[j] = obj[(size_t)j];
    try {
        return std::make_unique<Symmetrical>(A, b, c, maximize);
    } catch (const std::exception& e) {
        lastError_ = std::string("parse error: ") + e.what();
        return nullptr;
    }
}
