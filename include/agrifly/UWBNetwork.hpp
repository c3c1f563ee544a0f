// agrifly/UWBNetwork.hpp -- Simulation::UWBNetwork for hosts outside the agri-fly tree.
//
// Inside the tree keep the tree's own Components/Simulation/UWBNetwork.{hpp,cpp}: it works on the
// Simulation::UWBRadio objects agrifly::Quadcopter_T hands out through GetRadio() unchanged.  This
// header is the same class (same constructor, AddRadio, SetNoiseProperties, Run; the state machine
// of UWBNetwork.cpp:22-89 statement by statement) for standalone hosts and tests, with the noise
// drawn from the engine library's afe_uwb_network -- the same libstdc++ generator and distributions
// the reference instantiates at UWBNetwork.cpp:4-6, seeded 0 at construction (:19).  For ensembles,
// afe_uwb_range evaluates whole batches of transactions on the gathered positions on the GPU.
#pragma once
#ifdef AGRIFLY_USE_REFERENCE_TYPES
#include "Components/Simulation/UWBNetwork.hpp"
#else
#include <assert.h>

#include <memory>
#include <vector>

#include "SimulationObject6DOF.hpp"

namespace Simulation {

class UWBNetwork : public SimulationObject {
 public:
  UWBNetwork(BaseTimer *const masterTimer, double communicationPeriod)
      : SimulationObject(masterTimer), _commPeriod(communicationPeriod), _timeSinceLastRange(masterTimer),
        _currentRangingRequester(0), _currentRangingResponder(0), _addNoiseStdDev(0), _outlierProbability(0),
        _outlierStdDev(0), _net(0) {
    afe_uwb_create(&_net);   // rng.seed(0), UWBNetwork.cpp:19
  }
  virtual ~UWBNetwork() { afe_uwb_destroy(_net); }
  UWBNetwork(const UWBNetwork &) = delete;
  UWBNetwork &operator=(const UWBNetwork &) = delete;

  void AddRadio(std::shared_ptr<UWBRadio> r) { _radios.push_back(r); }

  void SetNoiseProperties(double noiseStdDev, double outlierProbability, double outlierStdDev) {
    _addNoiseStdDev = noiseStdDev;
    _outlierProbability = outlierProbability;
    _outlierStdDev = outlierStdDev;
    afe_uwb_set_noise(_net, noiseStdDev, outlierProbability, outlierStdDev);
  }

  virtual void Run() {
    for (auto radio = _radios.begin(); radio != _radios.end(); radio++) (*radio)->Run();   // :24-26
    if (_timeSinceLastRange.GetSeconds<double>() < _commPeriod) return;                      // :28-30
    if (!_currentRangingRequester || !_currentRangingResponder) {                            // :32-46
      for (auto radio = _radios.begin(); radio != _radios.end(); radio++) {
        if ((*radio)->GetNextRangingTargetId()) {
          _currentRangingRequester = (*radio)->GetId();
          _currentRangingResponder = (*radio)->GetNextRangingTargetId();
          break;
        }
      }
      _timeSinceLastRange.Reset();
      return;
    }
    Vec3d reqTruePos, resTruePos;                                                            // :50-64
    bool haveRequester = false, haveResponder = false;
    UWBRadio::RangingMeasurement meas;
    for (auto radio = _radios.begin(); radio != _radios.end(); radio++) {
      if ((*radio)->GetId() == _currentRangingRequester) { reqTruePos = (*radio)->GetPosition(); haveRequester = true; }
      if ((*radio)->GetId() == _currentRangingResponder) { resTruePos = (*radio)->GetPosition(); haveResponder = true; }
    }
    if (haveRequester && haveResponder) {
      double noise = 0;
      uint8_t outlier = 0;
      afe_uwb_draw(_net, 1, &noise, &outlier);                                               // :67-70, the draws
      if (outlier) meas.range = (float)noise;                                                // :68
      else meas.range = (float)((reqTruePos - resTruePos).GetNorm2() + noise);               // :71
      meas.haveNew = true;
      meas.responderId = _currentRangingResponder;
      meas.failure = false;
      for (auto radio = _radios.begin(); radio != _radios.end(); radio++) (*radio)->SetMeasurement(meas);   // :77-80
    } else {
      assert(0);
    }
    _currentRangingRequester = 0;                                                            // :86-87
    _currentRangingResponder = 0;
  }

 private:
  std::vector<std::shared_ptr<UWBRadio> > _radios;
  double _commPeriod;
  Timer _timeSinceLastRange;
  uint8_t _currentRangingRequester;
  uint8_t _currentRangingResponder;
  double _addNoiseStdDev, _outlierProbability, _outlierStdDev;
  afe_uwb_network *_net;
};

}  // namespace Simulation
#endif
