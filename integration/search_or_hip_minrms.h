// search_or_hip_minrms.h -- the binding a minrms maintainer adds to put the HIP
// engine behind bin/minrms.  Compiles only against the minrms source tree.
//
// SearchOrHipMinrms IS-A SearchOrDyn3d: it overrides Search() and Refine()
// (bin/minrms/search_or.h:83-85,116) and fills the tables the unchanged
// WriteMSF()/WriteChimeraInfo() read (bin/minrms/search_or_dyn3d.h:293-310).
// Those tables are `private` in the reference; the one-line change a maintainer
// makes is `private:` -> `protected:` at search_or_dyn3d.h:228 (this file's .cc
// gets the same effect with -Dprivate=protected for its own translation unit).
#pragma once
#include <or_gen.h>
#include "search_or_dyn3d.h"

struct minrms_engine;

namespace minrms {

class SearchOrHipMinrms : public SearchOrDyn3d {
 public:
  SearchOrHipMinrms(SearchOrDyn3d::Settings const &s);
  ~SearchOrHipMinrms();
  void Search(OrientationGenerator const &og, long start, long stop);
  void Refine();

 private:
  void FetchTables();
  minrms_engine *engine;
  long refine_max_iters;
};

}  // namespace minrms
