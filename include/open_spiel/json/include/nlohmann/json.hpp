// forwards to the MI355X host mirror: `nlohmann::json` for sources written against the reference's struct API
// (open_spiel/spiel.h:235-299) is the mirror's small JSON value (open_spiel_amd/csrc/host/osg_json.h: parse, dump with
// sorted keys and no whitespace as nlohmann's dump() prints, typed access) — the nlohmann library itself is not part of
// this tree.
#ifndef OSG_INCLUDE_NLOHMANN_JSON_HPP_
#define OSG_INCLUDE_NLOHMANN_JSON_HPP_
#include "../../../../../open_spiel_amd/csrc/host/osg_json.h"
namespace nlohmann {
using json = open_spiel::hip::Json;
}  // namespace nlohmann
#endif  // OSG_INCLUDE_NLOHMANN_JSON_HPP_
