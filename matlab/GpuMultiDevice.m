% GpuMultiDevice.m
% Shim for libocs.so (include/ocs.h, ocs_multi_*): the batch axis of a call over the GPUs of one node.  The reference has
% no batch axis (tests/solve_test_problem.m:37 integrates one trajectory per call); here x0 is nStates x batch, u is
% nControls x (2N+1) x batch, and a call returns the arrays of the whole batch plus the RCCL-reduced statistics.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_multi_device.py).  See INTEGRATION.md.
classdef GpuMultiDevice < handle
   properties
      h          % libpointer to the ocs_multi handle
      devices    % HIP device ids, block k of a batch goes to devices(k)
   end
   methods
      function obj = GpuMultiDevice(devices)
         obj.devices = int32(devices(:)');  obj.h = libpointer('voidPtr');
         ocs_check(calllib('libocs', 'ocs_multi_create', obj.h, obj.devices, numel(devices)));
      end
      function delete(obj)
         calllib('libocs', 'ocs_multi_destroy', obj.h.Value);
      end
      function c = replicate(obj, make)
         % one handle object per device: make() runs with that device current (ocs_set_device)
         c = cell(1, numel(obj.devices));
         for k = 1:numel(obj.devices)
            ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(k)));  c{k} = make();
         end
         ocs_check(calllib('libocs', 'ocs_set_device', obj.devices(1)));
      end
      function [x, J, stats] = compute_states(obj, integs, probs, x0, u)      % RK4Integrator.m:28-56 per trajectory
         batch = size(u, 3);  nAug = size(x0, 1) + 1;  N = (size(u, 2) - 1) / 2;
         x = zeros(nAug, N + 1, batch);  J = zeros(batch, 1);  stats = zeros(4, 1);
         gh = cellfun(@(g) g.hnd.Value, integs, 'UniformOutput', false);  ph = cellfun(@(p) p.h.Value, probs, 'UniformOutput', false);
         [rc, ~, ~, ~, ~, x, J, stats] = calllib('libocs', 'ocs_multi_compute_states', obj.h.Value, [gh{:}], [ph{:}], ...
               batch, x0, u, x, J, stats);
         ocs_check(rc);
      end
      function [J, dJdv, stats] = nlp_objective(obj, integs, probs, ctrls, x0, v)   % single_shooting.m:137-150
         batch = size(v, 2);  J = zeros(batch, 1);  dJdv = zeros(size(v));  stats = zeros(4, 1);
         gh = cellfun(@(g) g.hnd.Value, integs, 'UniformOutput', false);  ph = cellfun(@(p) p.h.Value, probs, 'UniformOutput', false);
         ch = cellfun(@(c) c.h.Value, ctrls, 'UniformOutput', false);
         [rc, ~, ~, ~, ~, ~, ~, J, dJdv, stats] = calllib('libocs', 'ocs_multi_nlp_objective', obj.h.Value, [gh{:}], [ph{:}], ...
               [ch{:}], batch, x0, v, 0, [], J, dJdv, stats);
         ocs_check(rc);
      end
   end
end
