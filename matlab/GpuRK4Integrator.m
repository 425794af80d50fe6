% GpuRK4Integrator.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's Integrator/RK4Integrator.m and RK4InfiniteIntegrator.m (Integrator/Integrator.m:6-15).
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
classdef GpuRK4Integrator < Integrator
   properties
      t, h, nSTEPS
      nAug         % rows of x, remembered by compute_states like size(obj.xK,1) (RK4Integrator.m:61)
      hnd
   end
   methods
      function obj = GpuRK4Integrator(tspan, tspanExtra, uStar)   % RK4Integrator.m:16 / RK4InfiniteIntegrator.m:12
         obj.hnd = libpointer('voidPtrPtr');
         if nargin == 1
            ocs_check(calllib('libocs', 'ocs_rk4_create', obj.hnd, tspan, numel(tspan)));
         else
            ocs_check(calllib('libocs', 'ocs_rk4inf_create', obj.hnd, tspan, numel(tspan), ...
                              tspanExtra, numel(tspanExtra), uStar, numel(uStar)));
         end
         obj.nSTEPS = numel(tspan) - 1;
         obj.t = zeros(1, 2*obj.nSTEPS + 1);  obj.h = zeros(1, obj.nSTEPS);
         [~, ~, obj.t] = calllib('libocs', 'ocs_integrator_t', obj.hnd.Value, obj.t);
         [~, ~, obj.h] = calllib('libocs', 'ocs_integrator_h', obj.hnd.Value, obj.h);
      end
      function [x, J] = compute_states(obj, prob, x0, u)          % RK4Integrator.m:28
         batch = size(u, 3);
         obj.nAug = numel(x0)/batch + 1;
         x = zeros(obj.nAug, obj.nSTEPS + 1, batch);  J = zeros(batch, 1);
         [rc, ~, ~, ~, ~, x, J] = calllib('libocs', 'ocs_compute_states', obj.hnd.Value, ...
                                          prob.h.Value, batch, x0, u, x, J);
         ocs_check(rc);
      end
      function [lam, dJdu] = compute_adjoints(obj, prob, u, lamT)  % RK4Integrator.m:59
         batch = size(u, 3);
         lam = zeros(obj.nAug, obj.nSTEPS + 1, batch);  dJdu = zeros(size(u));
         if nargin < 4, lamT = []; end          % NULL -> default e_last (:63-66)
         [rc, ~, ~, ~, ~, lam, dJdu] = calllib('libocs', 'ocs_compute_adjoints', obj.hnd.Value, ...
                                               prob.h.Value, batch, u, lamT, lam, dJdu);
         ocs_check(rc);
      end
      function delete(obj), calllib('libocs', 'ocs_integrator_destroy', obj.hnd.Value); end
   end
end
