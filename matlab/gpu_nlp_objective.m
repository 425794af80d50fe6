% gpu_nlp_objective.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's the nested nlpObjective of functions/single_shooting.m:137-150.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
function [J, dJdv, x0] = gpu_nlp_objective(integrator, prob, control, x0, v, FreeInitStates)
   % one call instead of the four lines of nlpObjective (single_shooting.m:137-150); v may be (nV x batch)
   if nargin < 6, FreeInitStates = []; end
   batch = size(v, 2);  J = zeros(batch, 1);  dJdv = zeros(size(v));
   [rc, ~, ~, ~, x0, ~, ~, J, dJdv] = calllib('libocs', 'ocs_nlp_objective', integrator.hnd.Value, ...
         prob.h.Value, control.hnd.Value, batch, x0, v, numel(FreeInitStates), int32(FreeInitStates), J, dJdv);
   ocs_check(rc);
end
